"""every launch of the default forward with its eager device time (HIP events), slowest first (argv[2] = "order": program order)"""
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
lib = _abi.load()
stream = torch.cuda.current_stream()
sp = stream.cuda_stream
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in prog.calls]
acc = [0.0] * len(prog.calls)
for it in range(4):
    for (fn, args, name), (e0, e1) in zip(prog.calls, evs):
        e0.record(stream); fn(sp, *args[1:]); e1.record(stream)
    torch.cuda.synchronize()
    if it:
        for i, (e0, e1) in enumerate(evs):
            acc[i] += e0.elapsed_time(e1) / 3
rows = [(acc[i] * 1e3, prog.calls[i][2]) for i in range(len(acc))]
if not (len(sys.argv) > 2 and sys.argv[2] == "order"):
    rows.sort(reverse=True)
print("%d launches, %.2f ms sequential" % (len(rows), sum(r[0] for r in rows) / 1e3))
for us, name in rows:
    print("%8.1f us  %s" % (us, name))
