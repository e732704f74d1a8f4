"""CPU: the C-ABI library loads and exports every symbol include/checkerpose_hip.h declares (no compute calls
without a GPU); argument validation paths that return before any launch; host-side planner logic."""
import ctypes as C
import os
import re

import pytest
import torch

from checkerpose_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "checkerpose_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
        assert n in _abi.SIGNATURES, "ctypes binding missing for " + n
    assert sorted(_abi.SIGNATURES) == names


def test_version_strerror_align(lib):
    assert lib.cp_version() >= 100
    assert lib.cp_strerror(0) == b"ok"
    assert b"invalid" in lib.cp_strerror(-1)
    assert lib.cp_chan_align(_abi.CP_F32) == 4 and lib.cp_chan_align(_abi.CP_BF16) == 8
    # 18 channels padded to 20 (f32): K = 9*20 = 180 -> 12 chunks of 16; 2 tiles of 16 rows; 1 KiB per fragment
    assert lib.cp_packed_weight_bytes(_abi.CP_F32, 18, 20, 3, 3) == 2 * 12 * 1024
    assert lib.cp_packed_weight_bytes(_abi.CP_BF16, 256, 768, 3, 3) == 16 * 216 * 1024


def test_argument_validation_returns_before_launch(lib):
    d = _abi.CpConvDesc()
    assert lib.cp_conv2d_igemm(None, C.byref(d), None, None, None, None, None, None) == -1     # null pointers
    assert lib.cp_bits_decode(None, None, 0, None, None, None, None, None, 1, 1) == -1
    assert lib.cp_graph_launch(None, None) == -1
    assert lib.cp_graph_destroy(None) == 0
    with pytest.raises(RuntimeError, match="invalid"):
        _abi.check(-1, "x")


def test_modules_fail_loudly_without_gpu():
    from tests.common import build_net, det_image
    net = build_net(full=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(det_image(1), None)
    net.train()
    with pytest.raises(RuntimeError, match="train-mode"):
        net(det_image(1), None)
    with pytest.raises(RuntimeError, match="parameter containers"):
        net.init_net.img_backbone(det_image(1))


def test_state_dict_keys_match_reference_layout():
    """Key names/shapes the reference's module tree produces, checked on the drop-in."""
    from tests.common import build_net
    sd = build_net().state_dict()
    expect = {
        "init_net.conv1x1.weight": (512, 1024, 1, 1), "init_net.conv1x1.bias": (512,),
        "init_net.pre_query_block.1.conv.0.weight": (64, 128, 1, 1), "init_net.pre_query_block.0.conv.1.running_var": (64,),
        "init_net.mlp.weight": (7, 64), "up_net.0.0.weight": (1024, 256, 3, 3), "up_net.0.1.weight": (256,),
        "up_net.0.3.weight": (256, 256, 3, 3), "up_net.0.6.weight": (256, 256, 3, 3), "up_net.0.7.bias": (256,),
        "up_net.1.1.weight": (256, 768, 3, 3), "up_net.2.1.weight": (256, 512, 3, 3), "up_net.2.5.running_mean": (256,),
        "refine_net.0.local_feat_ext_block.patch_generator.weight": (64, 256, 2, 2),
        "refine_net.0.pre_graph_module.0.weight": (256, 320), "refine_net.1.pre_graph_module.0.weight": (256, 512),
        "refine_net.2.pre_graph_module.2.bias": (256,), "refine_net.2.pre_query_block.2.conv.0.weight": (256, 512, 1, 1),
        "refine_net.1.query_block.mlps.0.weight": (256, 256), "refine_net.1.query_block.mlps.2.weight": (64, 256),
        "refine_net.1.query_block.mlps.4.weight": (2, 64), "seg_block.weight": (2, 256, 1, 1),
        "init_net.img_backbone.conv1.weight": (64, 3, 3, 3), "init_net.img_backbone.layer1.0.downsample.0.weight": (256, 64, 1, 1),
        "init_net.img_backbone.transition1.1.0.0.weight": (36, 256, 3, 3),
        "init_net.img_backbone.stage4.2.fuse_layers.3.0.2.0.weight": (144, 18, 3, 3),
        "init_net.img_backbone.stage3.0.fuse_layers.0.2.0.weight": (18, 72, 1, 1),
        "init_net.img_backbone.incre_modules.3.0.conv3.weight": (1024, 256, 1, 1),
    }
    for k, shp in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert not any("knn_idx" in k or "batch_indices" in k for k in sd)     # plain attributes, not buffers
    head = sum(v.numel() for k, v in sd.items() if "img_backbone" not in k and "num_batches" not in k)
    assert abs(head - 10.39e6) < 0.05e6      # head params 10.39 M at N=512


def test_common_ops_helpers():
    from checkerpose_amd.common_ops import from_dim_str_to_tuple, get_batch_size
    assert get_batch_size(0.75, 32) == (8, 24)
    assert from_dim_str_to_tuple("1024_256_32") == (1024, 256, 32) and from_dim_str_to_tuple(None) is None
