"""Is the training step GPU-bound or host-bound?  Times the HOST side of every phase of bench_train.py's step (no synchronisation
inside a step: how long the CPU needs to enqueue it) next to the synchronised step time.  If the host total approaches the step
time, fewer launches / graph nodes pay twice; if it is far below, only kernel time counts."""
import sys
import time
import torch
sys.path.insert(0, ".")
from checkerpose_amd.losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
from checkerpose_amd.losses.mask_loss import MaskLoss_interpolate
from checkerpose_amd.synthetic import build_net, det_image, det_tensor

dev = torch.device("cuda:0")
B, N = 32, 512
net = build_net(npoint=N, seed=1).to(dev).train()
net.set_compute_dtype("bf16")
img = det_image(B, seed=100).to(dev)
roi_gt = (det_tensor("t_roi", (B, 1, N), seed=0) > -0.5).float().to(dev)
x_gt = (det_tensor("t_x", (B, 16, N), seed=0) > 0).float().to(dev)
y_gt = (det_tensor("t_y", (B, 16, N), seed=0) > 0).float().to(dev)
m_vis = (det_tensor("t_mv", (B, 128, 128), seed=0) > 0).float().to(dev)
m_full = (det_tensor("t_mf", (B, 128, 128), seed=0) > -0.3).float().to(dev)
roi_loss, bit_loss, seg_loss = UnmaskedCodeLoss("BCE"), MaskedCodeLoss("BCE"), MaskLoss_interpolate()
import os
from checkerpose_amd.optim import Adam
opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True) if os.environ.get("CHECKERPOSE_BENCH_TORCH_ADAM") == "1" else Adam(net.parameters(), lr=2e-4)
p3d = torch.zeros(1, 3, N, device=dev).expand(B, -1, -1)
T = {k: 0.0 for k in ("zero_grad", "forward", "loss", "backward", "opt")}


def step(rec):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    roi, xb, yb, seg, _, _ = net(img, p3d, 3); t.append(time.perf_counter())
    nb = xb.shape[1]
    loss = roi_loss(roi, roi_gt) + bit_loss(xb, x_gt[:, :nb], roi_gt) + bit_loss(yb, y_gt[:, :nb], roi_gt) \
        + seg_loss(seg[:, 0:1], m_vis) + seg_loss(seg[:, 1:2], m_full); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if rec:
        for k, a, b in zip(T, t, t[1:]):
            T[k] += b - a
    return loss if os.environ.get("KEEP_LOSS") == "1" else None


for _ in range(5):
    l0 = step(False)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    l1 = step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("per step: synchronised %.2f ms, host enqueue %.2f ms  |  " % (total / n * 1e3, host / n * 1e3) +
      ", ".join("%s %.2f" % (k, v / n * 1e3) for k, v in T.items()), flush=True)
