#!/usr/bin/env python3
"""Golden vectors for the training-side ops, produced by the REFERENCE's own modules with
torch autograd.  Runs ONLY in the build container (needs /root/reference); only the small .npz is committed.

  python tests/golden/make_golden_train.py

Pinned: UnmaskedCodeLoss / MaskedCodeLoss (BCE and L1), MaskLoss_interpolate -- value and d/dlogits;
StaticGraph_module (eval BN) output and d/dx; Index2Feat_module gathers * mask -- d/dpatches.
Inputs are closed-form (checkerpose_amd/detweights.py:det_tensor) so the tests regenerate them from seeds.
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

from checkerpose_amd.detweights import det_tensor, fill_state_dict_  # noqa: E402
from tests import train_cases as TC  # noqa: E402

timm = types.ModuleType("timm")                     # backbone.py imports timm at module top; never called here
timm.create_model = lambda **kw: nn.Identity()
timm.models = types.SimpleNamespace(list_modules=lambda: [])
sys.modules["timm"] = timm

from losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss  # noqa: E402
from losses.mask_loss import MaskLoss_interpolate  # noqa: E402
from model import init as R_init, pipeline as R_pipe  # noqa: E402

out = {}

# ---- losses
for name, c in TC.CODE_CASES.items():
    pred, gt, mask = TC.code_inputs(c)
    p = pred.clone().requires_grad_(True)
    if c["masked"]:
        loss = MaskedCodeLoss(c["type"])(p, gt, mask)
    else:
        loss = UnmaskedCodeLoss(c["type"])(p, gt)
    loss.backward()
    out["code_%s_loss" % name] = loss.detach().numpy()
    out["code_%s_grad" % name] = p.grad.numpy()

for name, c in TC.MASK_CASES.items():
    pred, gt = TC.mask_inputs(c)
    p = pred.clone().requires_grad_(True)
    loss = MaskLoss_interpolate()(p[:, c["ch"]:c["ch"] + 1], gt)
    loss.backward()
    out["mask_%s_loss" % name] = loss.detach().numpy()
    out["mask_%s_grad" % name] = p.grad.numpy()

# ---- StaticGraph_module: output and d/dx under a seeded upstream gradient
for name, c in TC.EDGE_CASES.items():
    x, idx, gup = TC.edge_inputs(c)
    B = x.shape[0]
    mod = R_init.StaticGraph_module(c["Cin"], c["Cout"], idx[None].expand(B, -1, -1), leaky_slope=c["slope"])
    sd = mod.state_dict()
    fill_state_dict_(sd, c["seed"] + 7)
    mod.load_state_dict(sd)
    mod.eval()
    xr = x.clone().requires_grad_(True)
    bi = torch.arange(B)[:, None].expand(B, idx.shape[0] * idx.shape[1])
    y = mod(xr, bi)                                  # (B, Cout, N)
    y.backward(gup)
    out["edge_%s_out" % name] = y.detach().numpy()
    out["edge_%s_dx" % name] = xr.grad.numpy()

# ---- Index2Feat_module: d/dpatches of (gathers * mask) under a seeded upstream gradient
for name, c in TC.I2F_CASES.items():
    patches, x_id, y_id, mask, gup = TC.i2f_inputs(c)
    B, E, Hp, Wp = patches.shape
    k = c["k"]
    pr = patches.clone().requires_grad_(True)
    bi = torch.arange(B)[:, None].expand(B, x_id.shape[1])
    # pipeline.py:158-162 (the reference's own indexing expressions on the patch map), then :280's mask multiply
    sf1 = pr[bi, :, 2 * y_id, 2 * x_id]
    sf2 = pr[bi, :, 2 * y_id + k, 2 * x_id]
    sf3 = pr[bi, :, 2 * y_id, 2 * x_id + k]
    sf4 = pr[bi, :, 2 * y_id + k, 2 * x_id + k]
    local = torch.cat([sf1, sf2, sf3, sf4], dim=2) * mask[:, :, None]
    local.backward(gup)
    out["i2f_%s_out" % name] = local.detach().numpy()
    out["i2f_%s_dpatches" % name] = pr.grad.numpy()  # NCHW

np.savez_compressed(os.path.join(HERE, "train_ops.npz"), **out)
print("wrote train_ops.npz:", {k: v.shape for k, v in out.items()})
