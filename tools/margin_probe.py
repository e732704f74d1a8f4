#!/usr/bin/env python3
"""Margin-aware bf16 agreement (checkerpose_amd/agreement.py) of the timed bf16 path against the fp32 HIP path:
   python tools/margin_probe.py [--npoint 512|4096] [--lm] [--batch 8]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from checkerpose_amd.agreement import logit_agreement          # noqa: E402
from checkerpose_amd.synthetic import LM_OBJ_IDS, build_net, det_image       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--npoint", type=int, default=512)
ap.add_argument("--lm", action="store_true")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--selection", default="auto")
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
N, B = a.npoint, a.batch
net32 = build_net(npoint=N, seed=1, lm=a.lm).to(dev).set_compute_dtype("fp32")
net16 = build_net(npoint=N, seed=1, lm=a.lm).to(dev).set_compute_dtype("bf16").set_kernel_selection(a.selection)
img = det_image(B, seed=3).to(dev)
obj = torch.tensor([LM_OBJ_IDS[i % 13] for i in range(B)], device=dev) if a.lm else None
args = (img, None, obj) if a.lm else (img, None)
ref = net32(*args)
t = torch.zeros(B, 13, N, device=dev)
t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
kw = dict(obj_ids=obj) if a.lm else {}
forced = logit_agreement(net16.forward_teacher_forced(img, t, **kw), ref)
free = logit_agreement(net16(*args), ref, tau=forced["tau"], explain=True, knn_idx=net16.init_net.knn_idx,
                       graph_ids=(obj - 1) if a.lm else None)
keep = ("tau", "flips", "flips_above_margin", "max_flip_margin", "max_abs_dlogit", "mean_abs_dlogit", "bit_agreement_min_row", "xy_id_equal",
        "flip_rate_by_margin", "id_mismatches", "id_mismatches_explained", "id_mismatches_explained_frac", "id_mismatches_from_subtau_self_flip")
print(json.dumps({"config": vars(a), "teacher_forced": {k: forced[k] for k in keep if k in forced}, "free_running": {k: free[k] for k in keep if k in free}}))
# the worst teacher-forced flips: row, margin, dlogit
