"""Shared helpers for the parity tests: keypoints, module construction, deterministic fills, fixtures."""
import os

import numpy as np
import torch

from checkerpose_amd.detweights import det_image, det_tensor, fill_state_dict_  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def pc_normalize(pc):
    """reference aux_utils/pointnet2_utils.py:11-20"""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def p3d_from(xyz, n):
    return torch.as_tensor(pc_normalize(np.asarray(xyz[:n], dtype=np.float64).copy()), dtype=torch.float32).transpose(1, 0).unsqueeze(0)


def ape_p3d(n=512):
    return p3d_from(np.load(os.path.join(GOLDEN, "fps_lmo_obj01.npy")), n)


def lm_p3d(n=512):
    lm = np.load(os.path.join(GOLDEN, "fps_lm_15x1024.npy"))
    return torch.cat([p3d_from(lm[o].astype(np.float64), n) for o in range(15)], 0)   # (15,3,n)


def inject_feats(B, seed=0):
    """same closed form as tests/golden/make_golden.py"""
    return [det_tensor("feat%d" % i, (B, c, s, s), 6.0, seed).abs()
            for i, (c, s) in enumerate(zip((128, 256, 512, 1024), (64, 32, 16, 8)))]


def build_net(npoint=512, p3d=None, seed=0, lm=False, backbone="hrnet_w18", full=True):
    """The drop-in modules with the config of hr18GNN2_res6_gnn3Skip_mlpQuery(.txt), deterministic weights."""
    if lm:
        from checkerpose_amd.model.init_lm import InitNet_GNN
        from checkerpose_amd.model.pipeline_lm import PoseNet_GNNskip
    else:
        from checkerpose_amd.model.init import InitNet_GNN
        from checkerpose_amd.model.pipeline import PoseNet_GNNskip
    if p3d is None:
        p3d = lm_p3d(npoint) if lm else ape_p3d(npoint)
    init_net = InitNet_GNN(npoint=npoint, p3d_normed=p3d, res_log2=3, backbone_name=backbone, pretrain_backbone=False,
                           max_batch_size=8, num_graph_module=2, graph_k=20, graph_leaky_slope=0.2)
    if not full:
        fill_state_dict_(init_net.state_dict(), seed=seed)
        return init_net.eval()
    net = PoseNet_GNNskip(init_net=init_net, npoint=npoint, p3d_normed=p3d, res_log2=6, num_filters=256, max_batch_size=8,
                          query_dims=None, local_k=2, leaky_slope=0.01, num_graph_module=3, graph_k=20,
                          graph_leaky_slope=0.2, query_type="mlp")
    fill_state_dict_(net.state_dict(), seed=seed)
    return net.eval()


def oracle_kwargs():
    return dict(backbone="hrnet_w18", res_log2=6, init_n_graph=2, n_graph=3, local_k=2, slope=0.01, graph_slope=0.2,
                init_graph_slope=0.2)
