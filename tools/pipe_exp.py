#!/usr/bin/env python3
"""Experiment: two program instances alternating on two streams (cross-step pipelining) vs one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.common import build_net, det_image
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
nets = [build_net(seed=1).to(dev).set_compute_dtype("bf16") for _ in range(2)]
for n in nets: n.clone_outputs = False
img = det_image(B, seed=3).to(dev)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for n, s in zip(nets, streams):
    with torch.cuda.stream(s):
        for _ in range(3): n(img, None)
torch.cuda.synchronize()
def run(two, steps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        k = i % 2 if two else 0
        with torch.cuda.stream(streams[k]):
            nets[k](img, None)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)
for _ in range(2):
    print("B=%d one stream: %.0f crops/s   two alternating streams: %.0f crops/s" % (B, run(False), run(True)))
