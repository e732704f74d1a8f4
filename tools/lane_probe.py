import sys, time, os
import torch
sys.path.insert(0, ".")
from checkerpose_amd.synthetic import build_net, det_image
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for B in (1, 8, 32):
    for lanes in (True, False):
        net = build_net(seed=1).to(dev).set_compute_dtype("bf16")
        net.clone_outputs = False
        net.use_lanes = lanes
        img = det_image(B, seed=5).to(dev)
        for _ in range(3):
            net(img, None)
        buf = net.input_buffer(B); buf.copy_(img)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            net(buf, None)
        torch.cuda.synchronize()
        print("B=%d lanes=%s: %.3f ms" % (B, lanes, (time.perf_counter() - t0) / n * 1e3), flush=True)
        del net
