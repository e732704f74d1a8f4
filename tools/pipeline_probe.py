"""Does a free-running two-stream pipeline (two program instances, consecutive batches alternate between them, no join between
steps) beat back-to-back replays on one stream?  The tail of a step (refinement stages: one small kernel at a time) and the
head of the next (stem + layer1: bandwidth-bound) use different resources."""
import sys, time
import torch
sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.set_grad_enabled(False)
nets, bufs = [], []
for i in range(NS):
    n = build_net(512).to(dev).set_compute_dtype("bf16")
    n.clone_outputs = False
    img = det_image(B, seed=100 + i).to(dev)
    for _ in range(3):
        n(img, None)
    b = n.input_buffer(B); b.copy_(img)
    nets.append(n); bufs.append(b)
torch.cuda.synchronize()
K = 40


def run_single():
    t0 = time.perf_counter()
    for k in range(K):
        nets[k % NS](bufs[k % NS], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


streams = [torch.cuda.Stream(dev) for _ in range(NS)]


def run_pipe():
    t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k % NS]):
            nets[k % NS](bufs[k % NS], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(2):
    print("one stream : %.3f ms per batch of %d" % (run_single(), B), flush=True)
    print("%d streams  : %.3f ms per batch of %d" % (NS, run_pipe(), B), flush=True)
