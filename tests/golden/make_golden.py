#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs ONLY in the build container (needs /root/reference).  The reference's Python never
travels; only the small .npz/.npy outputs of this script are committed.

  python tests/golden/make_golden.py

What is pinned:
  * keypoints: the reference's FPS keypoint pickles (data files; .npy copies live in checkerpose_amd/data/)
  * knn index tables from reference `knn` (init.py:27)
  * per-block outputs of StaticGraph_module / Index2Feat_module / get_gdrn_upsample_module /
    Refine_moduleGNN / InitNet_GNN (features injected through the timm stub)
  * end-to-end PoseNet_GNNskip 6-tuples: (a) injected features, (b) oracle HRNet-W18 as the stubbed
    `timm` backbone (backbone arithmetic itself stays unpinned), (c) the LM twin with per-sample graphs.
    Round 5: every end-to-end fixture goes through center_and_repair (19 recorded bias values + the per-keypoint conv1x1 bias)
    so that its final ids are spatially diverse (>= 24 distinct x and y ids of 64) and EVERY logit has |z| >= 5e-4.
Run order when regenerating everything: make_golden.py, make_golden_r2.py (its n2 fixture reads e2e_injected), make_golden_r3.py,
make_golden_train.py, make_golden_trainstep.py, make_golden_n3.py.
Weights/inputs are closed-form (checkerpose_amd/detweights.py) so nothing large is stored.
"""
import os
import pickle
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "checkerpose"))

from checkerpose_amd.detweights import det_image, det_tensor, fill_state_dict_  # noqa: E402
from checkerpose_amd.model.backbone import HRNetW18Features  # noqa: E402
from oracle import checkerpose_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

# ---- timm stub: the reference's backbone.py does `import timm` at module top -----------------
_STUB = {"mode": "inject", "feats": None}


class _StubInject(nn.Module):
    """mode 'inject': ignore the image, return preset features (tests the reference HEAD alone)."""

    def forward(self, x):
        return [f.clone() for f in _STUB["feats"]]


class _StubHRNet(HRNetW18Features):
    """mode 'oracle_hrnet': parameters in timm naming (container from checkerpose_amd.model.backbone, so the
    state-dict keys are exactly `init_net.img_backbone.<timm key>`), forward = oracle restatement."""

    def forward(self, x):
        return O.hrnet_features(self.state_dict(), "", x)


def _StubBackbone(mode):
    return _StubInject() if mode == "inject" else _StubHRNet()


timm = types.ModuleType("timm")
timm.create_model = lambda **kw: _StubBackbone(_STUB["mode"])
timm.models = types.SimpleNamespace(list_modules=lambda: [])
sys.modules["timm"] = timm

from model import init as R_init, pipeline as R_pipe  # noqa: E402
from model import init_lm as R_init_lm, pipeline_lm as R_pipe_lm  # noqa: E402


def pc_normalize(pc):
    """aux_utils/pointnet2_utils.py:11-20"""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def load_fps(dataset, obj):
    with open("%s/checkerpose/datasets/BOP_DATASETS/%s/fps_202212/obj_%06d.pkl" % (REF, dataset, obj), "rb") as f:
        return pickle.load(f)["xyz"]


def p3d(xyz, n):
    return torch.as_tensor(pc_normalize(xyz[:n].copy()), dtype=torch.float32).transpose(1, 0).unsqueeze(0)


def inject_feats(B, seed=0):
    return [det_tensor("feat%d" % i, (B, c, s, s), 6.0, seed).abs()  # post-ReLU features are >= 0
            for i, (c, s) in enumerate(zip((128, 256, 512, 1024), (64, 32, 16, 8)))]


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def build_ref(npoint, p3d_normed, mode, seed, lm=False):
    _STUB["mode"] = mode
    I, P = (R_init_lm, R_pipe_lm) if lm else (R_init, R_pipe)
    init_net = I.InitNet_GNN(npoint=npoint, p3d_normed=p3d_normed, res_log2=3, backbone_name="hrnet_w18",
                             pretrain_backbone=False, max_batch_size=8, num_graph_module=2, graph_k=20,
                             graph_leaky_slope=0.2)
    net = P.PoseNet_GNNskip(init_net=init_net, npoint=npoint, p3d_normed=p3d_normed, res_log2=6, num_filters=256,
                            max_batch_size=8, query_dims=None, local_k=2, leaky_slope=0.01, num_graph_module=3,
                            graph_k=20, graph_leaky_slope=0.2, query_type="mlp")
    fill_state_dict_(net.state_dict(), seed=seed)
    net.eval()
    return net


def decision_margin(roi, xb, yb):
    """min |logit| over the logits that drive the discrete feedback (all but the last x/y bit)."""
    z = torch.cat([roi, xb[:, :-1], yb[:, :-1]], dim=1)
    return float(z.abs().min())


def all_logits(o):
    return torch.cat([o[0], o[1], o[2]], dim=1)                     # (B, 13, N): roi | x bits | y bits


def _midpoint(v):
    """value halfway between the two middle order statistics of v (an even count): subtracting it leaves half of v on either side
    and nothing AT zero"""
    s = torch.sort(v.reshape(-1)).values
    k = s.numel() // 2
    return 0.5 * (s[k - 1] + s[k])


def center_and_repair(net, fwd, target=1e-3, max_iter=400):
    """Round 5: fixtures whose ids cover the address space and whose EVERY logit has a margin.

    With the plain closed-form fill the 13 logit rows have a common sign over most keypoints (the post-ReLU features have a
    non-zero mean), so the final ids took 2-12 of 64 values and the stages' Index2Feat gathers ran at a handful of pixel sites.
    Two deterministic edits of the REFERENCE module's parameters, both recorded in the fixture (keys `ov__<state-dict key>`) so
    that the build's modules and the oracle apply the same values:
      1. centring -- `init_net.mlp.bias` (init.py:107) and each stage's `refine_net.i.query_block.mlps.4.bias` (pipeline.py:168-180)
         are shifted so that every logit row is split half / half over (batch, keypoint): stage by stage, because stage i's bits
         exist only once the earlier ids are fixed (pipeline.py:367-381);
      2. margin repair -- while any of the 13 x B x N logits has |z| < target, the offending keypoints' `init_net.conv1x1.bias[n]`
         (init.py:85-95: the only per-keypoint parameter; it shifts keypoint n's 64 graph features) is nudged by 0.004 x (times nudged so far), always the same way (enough to leave the band, too little to move the
         neighbours far); that
         re-rolls the logits of n and its graph neighbourhood and leaves the rest alone.
    Returns (outputs, overrides dict, margin over all 13 rows, iterations)."""
    sd = net.state_dict()                                           # shares storage with the module's parameters
    o = fwd()
    z = all_logits(o)
    rows7 = torch.cat([z[:, 0:4], z[:, 7:10]], 1)                   # InitNet's rows: roi, x2 x1 x0, y2 y1 y0 (pipeline.py:363-365)
    sd["init_net.mlp.bias"] -= torch.stack([_midpoint(rows7[:, r]) for r in range(7)])
    nstage = len(net.refine_net)
    for i in range(nstage):
        z = all_logits(fwd())
        sd["refine_net.%d.query_block.mlps.4.bias" % i] -= torch.stack([_midpoint(z[:, 4 + i]), _midpoint(z[:, 10 + i])])
    it = 0
    N = sd["init_net.conv1x1.bias"].numel()
    cnt, sgn = torch.zeros(N), torch.where(torch.arange(N) % 2 == 0, 1.0, -1.0)
    while True:
        o = fwd()
        z = all_logits(o)
        bad = (z.abs() < target).any(1).any(0)                      # (N,) keypoints with a logit inside the band, any sample / row
        nb = int(bad.sum())
        if nb == 0 or it >= max_iter:
            break
        cnt[bad] += 1                                               # a keypoint walks ONE way (by its parity), further each time
        sd["init_net.conv1x1.bias"][bad] += 0.004 * cnt[bad] * sgn[bad]
        it += 1
        print("   repair %3d: %d keypoints inside +-%.0e, min |z| %.2e" % (it, nb, target, float(z.abs().min())), flush=True)
    assert nb == 0, "margin repair did not converge"
    ov = {"ov__init_net.mlp.bias": sd["init_net.mlp.bias"].clone(), "ov__init_net.conv1x1.bias": sd["init_net.conv1x1.bias"].clone()}
    for i in range(nstage):
        k = "refine_net.%d.query_block.mlps.4.bias" % i
        ov["ov__" + k] = sd[k].clone()
    return o, ov, float(z.abs().min()), it


def id_diversity(o):
    return np.array([len(np.unique(o[4].numpy())), len(np.unique(o[5].numpy()))], dtype=np.int64)


def save_e2e(name, net, fwd, seed, extra=None):
    """one end-to-end fixture: the reference module's 6-tuple after center_and_repair, the parameter overrides, the margin over ALL
    13 logit rows (`margin`; the final ids need the last bits too), the number of distinct final x / y ids (`id_diversity`)"""
    o, ov, m, it = center_and_repair(net, fwd)
    div = id_diversity(o)
    frac = float((o[0] > 0).float().mean())
    print("%s: margin %.2e (all rows), distinct ids x %d / y %d of 64, roi-frac %.2f, %d repair rounds" % (name, m, div[0], div[1], frac, it))
    assert m >= 5e-4 and div.min() >= 24 and 0.15 < frac < 0.85
    more = extra() if extra else {}
    save(name, seed=seed, margin=m, id_diversity=div, roi=o[0], xb=o[1], yb=o[2], seg=o[3],
         xid=o[4].numpy().astype(np.int16), yid=o[5].numpy().astype(np.int16), **ov, **more)


def main():
    # ------------------------------------------------------------------ keypoints (data files)
    # (the keypoint .npy files themselves are written by make_golden_r2.py into checkerpose_amd/data/; LM keypoints are
    # stored as float32 there, so the LM goldens are made from the float32-rounded coordinates)
    ape = load_fps("lmo", 1)
    lm = np.stack([load_fps("lm", o)[:1024] for o in range(1, 16)]).astype(np.float32)

    # ------------------------------------------------------------------ knn tables
    for n in (512, 4096):
        idx = R_init.knn(p3d(ape, n), 20)
        assert bool((idx[0, :, 0] == torch.arange(n)).all()), "self is neighbour 0"
        save("knn_ape%d" % n, idx=idx.numpy().astype(np.int16))
    lm_p3d = torch.cat([p3d(lm[o].astype(np.float64), 512) for o in range(15)], 0)   # (15,3,512)
    save("knn_lm512", idx=R_init.knn(lm_p3d, 20).numpy().astype(np.int16))

    # ------------------------------------------------------------------ per-block, N=512, ape
    P512 = p3d(ape, 512)
    _STUB["feats"] = inject_feats(2)
    net = build_ref(512, P512, "inject", seed=0)
    sd = net.state_dict()
    B = 2
    bi_edge = net.init_net.pre_batch_indices[:B]
    x64 = det_tensor("x64", (B, 64, 512), 1.0)
    y64 = net.init_net.pre_query_block[0](x64, bi_edge)
    x256 = det_tensor("x256", (B, 256, 512), 1.0)
    y256 = net.refine_net[1].pre_query_block[2](x256, bi_edge)
    save("blk_edgeconv", y64=y64, y256=y256[:, :, ::4])

    for i, H in enumerate((16, 32, 64)):
        f = det_tensor("i2f%d" % H, (B, 256, H, H), 1.0)
        xid = torch.from_numpy((np.arange(B * 512).reshape(B, 512) * 7 + 3) % (H // 2)).long()
        yid = torch.from_numpy((np.arange(B * 512).reshape(B, 512) * 5 + 1) % (H // 2)).long()
        xid[:, :4] = torch.tensor([0, H // 2 - 1, 0, H // 2 - 1]); yid[:, :4] = torch.tensor([0, 0, H // 2 - 1, H // 2 - 1])
        out = net.refine_net[i].local_feat_ext_block(f, net.refine_net[i].batch_indices[:B], xid, yid)
        save("blk_index2feat_h%d" % H, out=out[:, :, ::4], xid=xid.numpy().astype(np.int16), yid=yid.numpy().astype(np.int16))

    up0 = net.up_net[0](det_tensor("up0", (1, 1024, 4, 4), 1.0).abs())
    up1 = net.up_net[1](det_tensor("up1", (1, 768, 6, 6), 1.0).abs())
    up2 = net.up_net[2](det_tensor("up2", (1, 512, 5, 7), 1.0).abs())
    save("blk_upsample", up0=up0, up1=up1, up2=up2)

    gfeat = det_tensor("gfeat", (B, 64, 512), 1.0)
    imf = det_tensor("imf16", (B, 256, 16, 16), 1.0).abs()
    roi = torch.where(det_tensor("roi", (B, 1, 512), 1.0) > -0.3, 1.0, 0.0)
    xid = torch.from_numpy((np.arange(B * 512).reshape(B, 512) * 3) % 8).long()
    yid = torch.from_numpy((np.arange(B * 512).reshape(B, 512) * 11 + 2) % 8).long()
    bits, gf = net.refine_net[0](imf, gfeat, P512.expand(B, -1, -1), roi, xid, yid)
    save("blk_refine0", bits=bits, feat=gf[:, :, ::4])

    out7, _, g0 = net.init_net(torch.zeros(B, 3, 256, 256), return_graph_feats=True)
    save("blk_initnet_injected", out=out7, graph=g0)

    # ------------------------------------------------------------------ end to end (a) injected features
    net = build_ref(512, P512, "inject", seed=0)
    save_e2e("e2e_injected", net, lambda: net(torch.zeros(B, 3, 256, 256), P512.expand(B, -1, -1)), seed=0)

    # ------------------------------------------------------------------ (b) oracle HRNet as the timm stub (B=1)
    img = det_image(1)
    net = build_ref(512, P512, "oracle_hrnet", seed=0)
    save_e2e("e2e_hrnet", net, lambda: net(img, P512), seed=0, extra=lambda: {"init_out": net.init_net(img)})   # init_out: config #1

    # ------------------------------------------------------------------ (c) LM twin, per-sample graphs
    obj_ids = torch.tensor([1, 9, 15])
    _STUB["feats"] = inject_feats(3, seed=1)
    net = build_ref(512, lm_p3d, "inject", seed=0, lm=True)
    save_e2e("e2e_lm_injected", net, lambda: net(torch.zeros(3, 3, 256, 256), lm_p3d[obj_ids - 1], obj_ids), seed=0,
             extra=lambda: {"obj_ids": obj_ids.numpy()})


if __name__ == "__main__":
    main()
