"""B = 1 / 8 forward: GPU-only span of one hipGraph replay (host enqueue hidden behind a sleep kernel) and back-to-back time for
max_lanes = 1, 2, 3, 4, all; plus the eager (one stream, no graph) time."""
import sys
import time
import torch
sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10000000 / e0.elapsed_time(e1)
for B in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 8)):
    for ml in ([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 3, 4, 0)):
        net = build_net(npoint=512, seed=1).to(dev).eval()
        net.set_compute_dtype("bf16")
        net.max_lanes = ml
        img = det_image(B, seed=3).to(dev)
        with torch.no_grad():
            for _ in range(3):
                net(img, None)
            buf = net.input_buffer(B)
            buf.copy_(img)
            for _ in range(10):
                net(buf, None)
            torch.cuda.synchronize()
            spans = []
            for _ in range(10):
                torch.cuda.synchronize()
                torch.cuda._sleep(int(3 * cyc_per_ms))
                e0.record()
                net(buf, None)
                e1.record()
                torch.cuda.synchronize()
                spans.append(e0.elapsed_time(e1))
            t0 = time.perf_counter()
            for _ in range(200):
                net(buf, None)
            torch.cuda.synchronize()
            bb = (time.perf_counter() - t0) / 200
        print("B=%d max_lanes=%d: GPU-only span min %.3f med %.3f ms; back to back %.3f ms" % (B, ml, min(spans), sorted(spans)[5], bb * 1e3), flush=True)
        del net
