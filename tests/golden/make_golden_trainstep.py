#!/usr/bin/env python3
"""Golden vectors of ONE TRAINING STEP of the REFERENCE's own modules in .train() mode (batch-statistics BatchNorm, torch
autograd), with the reference's own loss classes combined as in train.py:307-318.  Runs ONLY in the build container
(needs /root/reference); only the small .npz is committed.

  python tests/golden/make_golden_trainstep.py

The backbone is the injected-feature stub (the head is what the reference pins; the backbone arithmetic is timm's).
Stored: the five loss values, the 13 logit rows + seg logits, every gradient tensor of <= 4096 elements in full, and for
larger ones sum, abs-sum and 64 entries at fixed strides; running statistics of three BatchNorm layers after the step.
tests/test_oracle_train.py re-runs the oracle restatement (bn_train) on the same closed-form inputs against this file.
"""
import os
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

from checkerpose_amd.detweights import fill_state_dict_  # noqa: E402
from tests import train_cases as TC  # noqa: E402
from tests.common import ape_p3d, inject_feats  # noqa: E402

_FEATS = {}


class _StubInject(nn.Module):
    def forward(self, x):
        return [f.clone() for f in _FEATS["f"]]


timm = types.ModuleType("timm")
timm.create_model = lambda **kw: _StubInject()
timm.models = types.SimpleNamespace(list_modules=lambda: [])
sys.modules["timm"] = timm

from losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss  # noqa: E402
from losses.mask_loss import MaskLoss_interpolate  # noqa: E402
from model import init as R_init, pipeline as R_pipe  # noqa: E402


def main():
    c = TC.TRAINSTEP
    B, N = c["B"], c["N"]
    P = ape_p3d(N)
    _FEATS["f"] = inject_feats(B, seed=c["feat_seed"])
    init_net = R_init.InitNet_GNN(npoint=N, p3d_normed=P, res_log2=3, backbone_name="hrnet_w18", pretrain_backbone=False,
                                  max_batch_size=8, num_graph_module=2, graph_k=20, graph_leaky_slope=0.2)
    net = R_pipe.PoseNet_GNNskip(init_net=init_net, npoint=N, p3d_normed=P, res_log2=6, num_filters=256, max_batch_size=8,
                                 query_dims=None, local_k=2, leaky_slope=0.01, num_graph_module=3, graph_k=20,
                                 graph_leaky_slope=0.2, query_type="mlp")
    fill_state_dict_(net.state_dict(), seed=c["seed"])
    net.train()
    roi_gt, x_gt, y_gt, m_vis, m_full = TC.trainstep_targets(c)
    roi, xb, yb, seg, x_id, y_id = net(torch.zeros(B, 3, 256, 256), P.expand(B, -1, -1), None)
    nb = xb.shape[1]
    l_roi = UnmaskedCodeLoss("BCE")(roi, roi_gt)
    l_x = MaskedCodeLoss("BCE")(xb, x_gt[:, :nb], roi_gt)
    l_y = MaskedCodeLoss("BCE")(yb, y_gt[:, :nb], roi_gt)
    l_v = MaskLoss_interpolate()(seg[:, 0:1], m_vis)
    l_f = MaskLoss_interpolate()(seg[:, 1:2], m_full)
    loss = l_roi + l_x + l_y + l_v * c["w_vis"] + l_f * c["w_full"]          # train.py:317-318
    loss.backward()
    out = {"losses": np.array([float(v) for v in (l_roi, l_x, l_y, l_v, l_f, loss)], np.float64),
           "roi": roi.detach().numpy(), "xb": xb.detach().numpy(), "yb": yb.detach().numpy(), "seg": seg.detach().numpy(),
           "xid": x_id.numpy().astype(np.int16), "yid": y_id.numpy().astype(np.int16),
           "margin": float(torch.cat([roi, xb[:, :-1], yb[:, :-1]], 1).abs().min())}
    names = []
    for k, p in net.named_parameters():
        g = p.grad
        assert g is not None, k
        names.append(k)
        g = g.detach().reshape(-1)
        if g.numel() <= 4096:
            out["g:" + k] = g.numpy()
        else:
            idx = torch.arange(64) * (g.numel() // 64)
            out["g:" + k] = np.concatenate([[float(g.double().sum()), float(g.double().abs().sum())], g[idx].double().numpy()])
    sd = net.state_dict()
    for k in c["bn_probe"]:
        out["rm:" + k] = sd[k + ".running_mean"].numpy()
        out["rv:" + k] = sd[k + ".running_var"].numpy()
    path = os.path.join(HERE, "trainstep_injected.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB, margin %.2e, losses %s" % (path, os.path.getsize(path) / 1024, out["margin"], out["losses"]))


if __name__ == "__main__":
    main()
