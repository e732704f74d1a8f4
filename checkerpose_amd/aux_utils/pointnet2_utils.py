"""The one live symbol of reference checkerpose/aux_utils/pointnet2_utils.py: `pc_normalize`, the
keypoint normalisation every script applies before building the kNN graph (train.py / test.py: `pc_normalize(p3d_xyz)`).
Construction-time host prep (numpy, like the reference), not part of the per-crop hot path."""
import numpy as np


def pc_normalize(pc, return_stat=False):
    """pointnet2_utils.py:11-20: centre on the centroid, scale by the largest centred radius.  pc (N, 3)."""
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    pc = pc / m
    if return_stat:
        return pc, centroid, m
    return pc
