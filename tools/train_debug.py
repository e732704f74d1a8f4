"""Debug aid: per-parameter gradient error of one training step (HIP program vs oracle autograd), in network order."""
import sys
import torch
sys.path.insert(0, ".")
from tests.common import build_net, det_image, det_tensor
from tests.test_gpu_train_step import _oracle_step

stage = None
B = 2
net = build_net(seed=3).train()
img = det_image(B, seed=11)
active = 3
seeds = [det_tensor("g_roi", (B, 1, 512)), det_tensor("g_x", (B, 3 + active, 512)), det_tensor("g_y", (B, 3 + active, 512)),
         det_tensor("g_seg", (B, 2, 8 << active, 8 << active), 0.05)]
from tests.test_gpu_train_step import _device_kstar
net_cpu = build_net(seed=3).train()
net = net.cuda()
with torch.enable_grad():
    res = net(img.cuda(), None, stage)
    torch.autograd.backward(list(res[:4]), [s.cuda() for s in seeds])
torch.cuda.synchronize()
outs, ids, ref, sd_ref = _oracle_step(net_cpu, img, seeds, stage=stage, kstar=_device_kstar(net))
for k, p in net.named_parameters():
    g_ref, g = ref[k], p.grad
    if g_ref is None:
        print("%-70s ref None  got max %.3e" % (k, float(g.abs().max()) if g is not None else -1))
        continue
    sc = float(g_ref.abs().max())
    err = float((g.cpu() - g_ref).abs().max()) / max(sc, 1e-12)
    flag = "" if err < 2e-3 else "   <<<<<<"
    print("%-70s scale %.3e relerr %.3e%s" % (k, sc, err, flag))

