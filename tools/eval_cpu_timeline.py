"""Is the small-batch eval forward bound by the GPU or by the host's hipGraph launch?  Host time of `net(img)` (no synchronisation) next
to the synchronised time per forward, for a few batch sizes."""
import sys
import time
import torch
sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image

dev = torch.device("cuda:0")
net = build_net(npoint=512, seed=1).to(dev).eval()
net.set_compute_dtype("bf16")
for B in ([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 8, 32, 256)):
    img = det_image(B, seed=3).to(dev)
    with torch.no_grad():
        net(img, None)
    buf = net.input_buffer(B)
    buf.copy_(img)
    with torch.no_grad():
        for _ in range(20):
            net(buf, None)
        torch.cuda.synchronize()
        n = 300 if B <= 32 else 50
        t0 = time.perf_counter()
        for _ in range(n):
            net(buf, None)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        # one forward alone: launch, then wait
        lat = []
        for _ in range(50):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net(buf, None)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            lat.append((t2 - t1, time.perf_counter() - t1))
    prog = net.program_for(B)
    print("B=%3d: %4d launches | back to back: %.3f ms per forward synchronised, %.3f ms host enqueue | single forward: host %.3f ms, "
          "until done %.3f ms" % (B, len(prog.calls), total / n * 1e3, host / n * 1e3, sum(a for a, _ in lat) / 50 * 1e3,
                                  sum(b for _, b in lat) / 50 * 1e3), flush=True)
