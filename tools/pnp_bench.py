"""micro-benchmark of cp_pnp_ransac on synthetic poses (LM-O ape keypoints, 30 % outliers, 0.5 px noise): ms per batch"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from checkerpose_amd.postprocess import solve_pnp_ransac
from tests.test_pnp import K_LMO, make_case
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
cases = [make_case(rng) for _ in range(8)]
xyz = torch.from_numpy(cases[0][0]).float().to(dev)
p2d = torch.from_numpy(np.stack([cases[i % 8][1] for i in range(B)])).float().to(dev)
valid = torch.zeros(B, 512, 3, dtype=torch.uint8, device=dev)
valid[:, :, 0] = torch.from_numpy(np.stack([cases[i % 8][2] for i in range(B)])).to(dev)
K = torch.from_numpy(K_LMO).float().to(dev)
for it in (150, 50):
    run = lambda: solve_pnp_ransac(xyz, p2d, valid, K, iterations=it)
    for _ in range(2):
        R, t, inl, st = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print("B=%d iterations=%d: %.3f ms per batch, solved %d, mean inliers %.0f" % (B, it, e0.elapsed_time(e1) / 5, int(st.sum()), float(inl.sum(1).float().mean())))
