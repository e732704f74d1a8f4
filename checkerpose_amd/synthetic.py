"""Synthetic workloads for bench.py / smoke() / the tests: keypoint data, module construction with the shipped config
values (config/lmo/hr18GNN2_res6_gnn3Skip_mlpQuery.txt:14-28) and deterministic closed-form weights / crops.

Keypoint files under checkerpose_amd/data/ are DATA extracted from the reference's FPS pickles
(datasets/BOP_DATASETS/{lmo,lm,ycbv}/fps_202212/obj_*.pkl, key "xyz") by tests/golden/make_golden_r2.py:
  fps_lmo_obj01.npy    (4096,3) f64   LM-O `ape`                     (BASELINE configs #1-#3)
  fps_ycbv_21x512.npy  (21,512,3) f32 the 21 YCB-V objects           (config #4: one network per object, train.py:384,396)
  fps_lm_15x4096.npy   (15,4096,3) f32 the 15 LM objects             (config #5: shared estimator, kNN table (15,N,K))
"""
import os

import numpy as np
import torch

from .detweights import det_image, det_tensor, fill_state_dict_  # noqa: F401

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
LM_OBJ_IDS = (1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15)     # the 13 evaluated LM objects, test_network_with_test_data.py:533


def pc_normalize(pc):
    """reference aux_utils/pointnet2_utils.py:11-20"""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def p3d_from(xyz, n):
    """(>=n,3) keypoints -> p3d_normed (1,3,n) f32 as test.py:147-157 builds it"""
    return torch.as_tensor(pc_normalize(np.asarray(xyz[:n], dtype=np.float64).copy()), dtype=torch.float32).transpose(1, 0).unsqueeze(0)


def ape_p3d(n=512):
    return p3d_from(np.load(os.path.join(DATA, "fps_lmo_obj01.npy")), n)


def lm_p3d(n=512):
    lm = np.load(os.path.join(DATA, "fps_lm_15x4096.npy"))
    return torch.cat([p3d_from(lm[o].astype(np.float64), n) for o in range(15)], 0)   # (15,3,n)


def ycbv_p3d(obj, n=512):
    """obj: 1..21"""
    return p3d_from(np.load(os.path.join(DATA, "fps_ycbv_21x512.npy"))[obj - 1].astype(np.float64), n)


def apply_overrides(sd, overrides):
    """Parameter values a fixture recorded beside its seed (npz keys `ov__<state-dict key>`: tests/golden/make_golden.py
    center_and_repair) copied into a state dict in place; returns the number of tensors set."""
    n = 0
    if overrides is None:
        return n
    for k in getattr(overrides, "files", None) or overrides.keys():
        if k.startswith("ov__"):
            with torch.no_grad():
                sd[k[4:]].copy_(torch.as_tensor(np.asarray(overrides[k])))
            n += 1
    return n


def build_net(npoint=512, p3d=None, seed=0, lm=False, backbone="hrnet_w18", full=True, overrides=None, init_graph=2, graph=3):
    """The drop-in modules with the config of hr18GNN2_res6_gnn3Skip_mlpQuery(.txt), deterministic weights (+ a fixture's
    recorded parameter overrides)."""
    if lm:
        from .model.init_lm import InitNet_GNN
        from .model.pipeline_lm import PoseNet_GNNskip
    else:
        from .model.init import InitNet_GNN
        from .model.pipeline import PoseNet_GNNskip
    if p3d is None:
        p3d = lm_p3d(npoint) if lm else ape_p3d(npoint)
    init_net = InitNet_GNN(npoint=npoint, p3d_normed=p3d, res_log2=3, backbone_name=backbone, pretrain_backbone=False,
                           max_batch_size=8, num_graph_module=init_graph, graph_k=20, graph_leaky_slope=0.2)
    if not full:
        fill_state_dict_(init_net.state_dict(), seed=seed)
        return init_net.eval()
    net = PoseNet_GNNskip(init_net=init_net, npoint=npoint, p3d_normed=p3d, res_log2=6, num_filters=256, max_batch_size=8,
                          query_dims=None, local_k=2, leaky_slope=0.01, num_graph_module=graph, graph_k=20,
                          graph_leaky_slope=0.2, query_type="mlp")
    fill_state_dict_(net.state_dict(), seed=seed)
    apply_overrides(net.state_dict(), overrides)
    return net.eval()
