"""forward time vs batch for the two kernel selections (per-crop fused launches on / off)"""
import sys, time
import torch
sys.path.insert(0, ".")
from checkerpose_amd import engine
from checkerpose_amd.synthetic import build_net, det_image
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for B in (1, 8, 16, 32, 64, 96, 128, 192, 256):
    res = []
    for sel in ("percrop", "tiles", "stem+edge only", "chain only"):
        engine.CHAIN_MIN_BATCH = 1 if sel in ("percrop", "chain only") else 1 << 30
        engine.EDGE_FUSED_MIN_BATCH = 1 if sel in ("percrop", "stem+edge only") else 1 << 30
        engine.STEM_MIN_BATCH = 1 if sel in ("percrop", "stem+edge only") else 1 << 30
        net = build_net(seed=1).to(dev).set_compute_dtype("bf16")
        net.clone_outputs = False
        img = det_image(B, seed=5).to(dev)
        for _ in range(3):
            net(img, None)
        torch.cuda.synchronize()
        n = 30 if B <= 32 else 10
        t0 = time.perf_counter()
        for _ in range(n):
            net(img, None)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
        del net
        torch.cuda.empty_cache()
    print("B=%3d  percrop %.3f ms  tiles %.3f ms  stem+edge %.3f  chain %.3f" % (B, *res), flush=True)
