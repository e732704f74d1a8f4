"""Cross-step pipelining probe: consecutive steps (whole batches) of the default forward on S program instances / S HIP streams in
turn, no join between steps -- does the latency-bound keypoint-side tail of step i (one workgroup per CU) share the chip with the
HBM-bound head (stem, layer1) of step i + 1?  Alternates S = 1 and S = 2 (and 3) inside one process.

    python tools/cross_step_probe.py [B=256] [steps=60] [reps=4]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
img = det_image(B, seed=0).to(dev)
SMAX = 3
streams = [torch.cuda.Stream(dev) for _ in range(SMAX)]
nets, bufs = [], []
for k in range(SMAX):
    with torch.cuda.stream(streams[k]):               # a program replays on the stream it was built under
        n = build_net(512, seed=1).to(dev).set_compute_dtype("bf16")
        n.clone_outputs = False
        n(img, None)
        b = n.input_buffer(B)
        b.copy_(img)
        n(b, None)
    nets.append(n); bufs.append(b)
torch.cuda.synchronize()


def run(S, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % S
        with torch.cuda.stream(streams[k]):
            nets[k](bufs[k], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for S in (1, 2, 3):
    run(S, 10)
res = {1: [], 2: [], 3: []}
for r in range(REPS):
    for S in (1, 2, 3):
        res[S].append(run(S, STEPS))
for S, v in res.items():
    v = sorted(v)
    print("S=%d: ms per step min %.3f median %.3f max %.3f -> %.0f crops/s" % (S, v[0], v[len(v) // 2], v[-1], B / v[len(v) // 2] * 1e3), flush=True)
# results identical?
o1 = [t.clone() for t in nets[0](bufs[0], None)]
with torch.cuda.stream(streams[1]):
    o2 = [t.clone() for t in nets[1](bufs[1], None)]
torch.cuda.synchronize()
print("instances agree bitwise:", all(torch.equal(a, b) for a, b in zip(o1, o2)))
