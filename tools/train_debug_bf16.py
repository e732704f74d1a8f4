"""Debug aid: cosine between the fp32 and bf16 training programs' gradients, per parameter."""
import sys
import torch
sys.path.insert(0, ".")
from tests.common import build_net, det_image, det_tensor
B = 4
img = det_image(B, seed=7).cuda()
seeds = [det_tensor("g_roi", (B, 1, 512)).cuda(), det_tensor("g_x", (B, 6, 512)).cuda(), det_tensor("g_y", (B, 6, 512)).cuda(),
         det_tensor("g_seg", (B, 2, 64, 64), 0.05).cuda()]
grads, outs = {}, {}
for dt in ("fp32", "bf16"):
    net = build_net(seed=3).cuda().train()
    net.set_compute_dtype(dt)
    with torch.enable_grad():
        res = net(img, None)
        torch.autograd.backward(list(res[:4]), seeds)
    grads[dt] = {k: p.grad.clone() for k, p in net.named_parameters()}
    outs[dt] = [r.detach().clone() for r in res]
torch.cuda.synchronize()
for i, nm in enumerate(("roi", "xb", "yb", "seg")):
    print(nm, "max diff", float((outs["fp32"][i] - outs["bf16"][i]).abs().max()), "bit agreement",
          float(((outs["fp32"][i] > 0) == (outs["bf16"][i] > 0)).float().mean()))
print("ids equal frac", float((outs["fp32"][4] == outs["bf16"][4]).float().mean()), float((outs["fp32"][5] == outs["bf16"][5]).float().mean()))
for k, g in grads["fp32"].items():
    a, b = g.flatten().double(), grads["bf16"][k].flatten().double()
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print("%-70s cos %.4f  |f32| %.3e |bf16| %.3e %s" % (k, cos, float(a.norm()), float(b.norm()), "" if cos > 0.9 else "  <<<<"))
