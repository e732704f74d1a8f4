"""probe: does a per-crop LDS-heavy launch (edge_fused) get CUs while a decoder conv (halo4, 1 wave/SIMD, all registers) runs on
another stream?  Prints edge_fused's latency alone / beside the conv, and the conv's time alone / beside."""
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd.synthetic import build_net, det_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = build_net(512).to(dev).set_compute_dtype("bf16")
net.clone_outputs = False
img = det_image(B, seed=100).to(dev)
for _ in range(2):
    net(img, None)
prog = net.program_for(B)
calls = {c[2]: c for c in prog.calls}
big = calls["conv3x3_halo4:up_net.2.1"]
small = [c for c in prog.calls if c[2].startswith("edge_fused:refine_net.1")][0]
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=prio)      # priority -1 = high (lower number = higher priority)


def run(c, s):
    rc = c[0](s.cuda_stream, *c[1][1:])
    assert rc == 0


def timed(fn):
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        out.append(fn())
    return sorted(out)[len(out) // 2]


def alone(c, s):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); run(c, s); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def both(nsmall):
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record(sa); run(big, sa); a1.record(sa)
    b0.record(sb)
    for _ in range(nsmall):
        run(small, sb)
    b1.record(sb)
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) * 1e3, b0.elapsed_time(b1) * 1e3, a0.elapsed_time(b1) * 1e3


print("conv alone %.1f us, edge_fused alone %.1f us" % (timed(lambda: alone(big, sa)), timed(lambda: alone(small, sb))))
for n in (1, 4, 10):
    r = [both(n) for _ in range(5)][2:]
    print("beside (x%d): conv %.1f us, %d edge_fused %.1f us, span %.1f us" % ((n,) + tuple(sum(x[i] for x in r) / len(r) for i in (0,)) + (n,) + tuple(sum(x[i] for x in r) / len(r) for i in (1, 2))))
