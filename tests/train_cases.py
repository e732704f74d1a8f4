"""Seeded inputs of the training-side parity cases -- shared by tests/golden/make_golden_train.py (reference run,
build container) and the tests (oracle + HIP), so only the expected outputs are stored."""
import torch

from checkerpose_amd.detweights import det_tensor as _det


def det_tensor(shape, seed):
    return _det("train_case", shape, 1.0, seed)

CODE_CASES = {
    "roi_bce": dict(B=3, nb=1, N=40, gt_rows=1, masked=False, type="BCE", seed=11),
    "roi_l1": dict(B=3, nb=1, N=40, gt_rows=1, masked=False, type="L1", seed=12),
    "proj_bce": dict(B=3, nb=4, N=40, gt_rows=16, masked=True, type="BCE", seed=13),     # gt sliced [:, :nb] like train.py:312
    "proj_l1": dict(B=3, nb=6, N=40, gt_rows=16, masked=True, type="L1", seed=14),
    "proj_bce_empty": dict(B=2, nb=3, N=24, gt_rows=16, masked=True, type="BCE", seed=15, empty_mask=True),  # clamp(min=1)
}
MASK_CASES = {
    "visib_128": dict(B=3, h=64, w=64, Hm=128, Wm=128, ch=0, seed=21),
    "full_64": dict(B=2, h=64, w=64, Hm=64, Wm=64, ch=1, seed=22),
    "odd_100": dict(B=2, h=64, w=64, Hm=100, Wm=90, ch=0, seed=23),
}
EDGE_CASES = {
    "c32": dict(B=3, Cin=16, Cout=32, N=48, K=6, slope=0.2, seed=31),
    "c64": dict(B=2, Cin=64, Cout=64, N=64, K=20, slope=0.1, seed=32),
}
I2F_CASES = {
    "k2": dict(B=3, E=8, H=16, N=40, k=2, seed=41),     # low-res ids in [0, H/2): many collisions
}


def _binary(shape, seed):
    return (det_tensor(shape, seed) > 0).float()


def code_inputs(c):
    pred = det_tensor((c["B"], c["nb"], c["N"]), c["seed"]) * 4.0
    gt_full = _binary((c["B"], c["gt_rows"], c["N"]), c["seed"] + 1)
    gt = gt_full[:, :c["nb"]]                       # non-contiguous view, as in train.py:312-313
    mask = None
    if c["masked"]:
        mask = _binary((c["B"], 1, c["N"]), c["seed"] + 2)
        if c.get("empty_mask"):
            mask = torch.zeros_like(mask)
    return pred, gt, mask


def mask_inputs(c):
    pred = det_tensor((c["B"], 2, c["h"], c["w"]), c["seed"]) * 3.0
    gt = _binary((c["B"], c["Hm"], c["Wm"]), c["seed"] + 1)
    return pred, gt


def edge_inputs(c):
    x = det_tensor((c["B"], c["Cin"], c["N"]), c["seed"])
    pts = det_tensor((1, 3, c["N"]), c["seed"] + 1)
    d = -((pts[:, :, :, None] - pts[:, :, None, :]) ** 2).sum(1)            # (1,N,N)
    idx = d.topk(k=c["K"], dim=-1)[1][0]                                    # (N,K) int64, self included
    gup = det_tensor((c["B"], c["Cout"], c["N"]), c["seed"] + 2)
    return x, idx, gup


def i2f_inputs(c):
    B, E, H, N, k = c["B"], c["E"], c["H"], c["N"], c["k"]
    Hp = H + k - 1                                                          # Conv2d(k, stride 1, padding k-1)
    patches = det_tensor((B, E, Hp, Hp), c["seed"])
    u = (det_tensor((2, B, N), c["seed"] + 1) * 0.5 + 0.5).clamp(0, 0.999)
    x_id = (u[0] * (H // 2)).long()
    y_id = (u[1] * (H // 2)).long()
    mask = _binary((B, N), c["seed"] + 2)
    gup = det_tensor((B, N, 4 * E), c["seed"] + 3)
    return patches, x_id, y_id, mask, gup


def edge_folded_weights(c):
    """StaticGraph_module's conv+BN (deterministic fill, same seed as the golden run) folded into the factored form
    the kernels use: wpq (2C', Cin) = [W1 ; W2 - W1], scale (2C') = [s ; s], shift (2C') = [0 ; t]
    (checkerpose_amd/netbuilder.py:NetEmitter.edgeconv)."""
    import torch.nn as nn
    from checkerpose_amd.detweights import fill_state_dict_

    class _M(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Sequential(nn.Conv2d(2 * c["Cin"], c["Cout"], 1, bias=False), nn.BatchNorm2d(c["Cout"]),
                                      nn.LeakyReLU(c["slope"]))
    sd = _M().state_dict()
    fill_state_dict_(sd, c["seed"] + 7)
    w = sd["conv.0.weight"][:, :, 0, 0]
    w1, w2 = w[:, :c["Cin"]], w[:, c["Cin"]:]
    s = sd["conv.1.weight"] / torch.sqrt(sd["conv.1.running_var"] + 1e-5)
    t = sd["conv.1.bias"] - sd["conv.1.running_mean"] * s
    return torch.cat([w1, w2 - w1], 0), torch.cat([s, s]), torch.cat([torch.zeros_like(t), t])


# ---- one full training step of the head (reference modules in .train() mode), tests/golden/make_golden_trainstep.py
TRAINSTEP = dict(B=2, N=512, seed=4, feat_seed=1, w_vis=1.0, w_full=1.0,
                 bn_probe=("up_net.0.1", "up_net.2.5", "refine_net.1.pre_query_block.0.conv.1", "init_net.pre_query_block.1.conv.1"))


def trainstep_targets(c):
    B, N = c["B"], c["N"]
    roi_gt = (det_tensor((B, 1, N), 51) > -0.5).float()
    x_gt = _binary((B, 16, N), 52)
    y_gt = _binary((B, 16, N), 53)
    m_vis = _binary((B, 128, 128), 54)
    m_full = (det_tensor((B, 128, 128), 55) > -0.3).float()
    return roi_gt, x_gt, y_gt, m_vis, m_full


# MaskedCodeLoss("CE") cases of tests/golden/make_golden_r3.py: name -> (B, C, N, seed, empty mask)
CE_CASES = {"c8": (3, 8, 40, 31, False), "c64": (2, 64, 24, 32, False), "empty": (2, 5, 16, 33, True)}


def ce_inputs(B, C, N, seed, empty=False):
    """same closed form as make_golden_r3.py:ce_inputs"""
    import torch
    from checkerpose_amd.detweights import det_tensor as _dt
    pred = _dt("ce_pred", (B, C, N), 3.0, seed)
    gt = (_dt("ce_gt", (B, 1, N), 1.0, seed).abs() * 1e4).long() % C
    mask = torch.zeros(B, 1, N) if empty else (_dt("ce_mask", (B, 1, N), 1.0, seed) > -0.2).float()
    return pred, gt, mask
