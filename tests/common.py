"""Shared helpers for the parity tests: keypoints, module construction, deterministic fills, fixtures."""
import os

import numpy as np
import torch  # noqa: F401

from checkerpose_amd.detweights import det_image, det_tensor, fill_state_dict_  # noqa: F401
from checkerpose_amd.synthetic import DATA, LM_OBJ_IDS, ape_p3d, build_net, lm_p3d, p3d_from, pc_normalize, ycbv_p3d  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def inject_feats(B, seed=0):
    """same closed form as tests/golden/make_golden.py"""
    return [det_tensor("feat%d" % i, (B, c, s, s), 6.0, seed).abs()
            for i, (c, s) in enumerate(zip((128, 256, 512, 1024), (64, 32, 16, 8)))]


def oracle_kwargs():
    return dict(backbone="hrnet_w18", res_log2=6, init_n_graph=2, n_graph=3, local_k=2, slope=0.01, graph_slope=0.2,
                init_graph_slope=0.2)
