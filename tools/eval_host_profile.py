"""cProfile of the HOST side of the B = 1 eval forward (where do the ~1.5 ms per forward go: hipGraphLaunch or Python?)"""
import cProfile
import pstats
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image

dev = torch.device("cuda:0")
net = build_net(npoint=512, seed=1).to(dev).eval()
net.set_compute_dtype("bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = det_image(B, seed=3).to(dev)
with torch.no_grad():
    for _ in range(20):
        net(img, None)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        net(img, None)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
