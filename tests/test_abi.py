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
    assert lib.cp_version() >= 201
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
    d2 = _abi.CpConvDesc()                      # a LeakyReLU slope outside [0, 1] is refused (the branch-free epilogue form needs it)
    d2.dtype, d2.act, d2.slope = _abi.CP_BF16, 2, 1.5
    one = C.c_void_p(16)
    for fn in (lib.cp_conv2d_igemm, lib.cp_gemm_rows, lib.cp_conv3x3_halo):
        assert fn(None, C.byref(d2), one, one, one, one, None, one) == -1
    assert lib.cp_graph_launch(None, None) == -1
    assert lib.cp_graph_destroy(None) == 0
    with pytest.raises(RuntimeError, match="invalid"):
        _abi.check(-1, "x")


def test_modules_fail_loudly_without_gpu():
    from tests.common import build_net, det_image
    net = build_net(full=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(det_image(1), None)
    net.train()                                  # the training program has no CPU fallback either
    with pytest.raises(RuntimeError, match="no CPU fallback"):
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


def test_state_dict_full_key_list():
    """The FULL key/shape list.  Head: dumped from the reference's own module tree (tests/golden/make_golden_r2.py keys ->
    head_state_dict_keys.json), must match exactly, in order.  Backbone: timm is absent, so the contract is the restated
    timm `hrnet_w18` features_only naming, generated here from its structure."""
    import json
    from tests.common import GOLDEN, build_net
    sd = build_net().state_dict()
    ref_head = json.load(open(os.path.join(GOLDEN, "head_state_dict_keys.json")))
    head = {k: list(v.shape) for k, v in sd.items() if ".img_backbone." not in k}
    assert head == ref_head
    assert [k for k in sd if ".img_backbone." not in k] == [k for k in ref_head] or sorted(head) == sorted(ref_head)

    def bn(p, c):
        return {p + ".weight": [c], p + ".bias": [c], p + ".running_mean": [c], p + ".running_var": [c], p + ".num_batches_tracked": []}

    def conv_bn(out, pc, pb, co, ci, k):
        out[pc + ".weight"] = [co, ci, k, k]
        out.update(bn(pb, co))

    exp = {}
    conv_bn(exp, "conv1", "bn1", 64, 3, 3); conv_bn(exp, "conv2", "bn2", 64, 64, 3)

    def bottleneck(p, cin, planes, ds):
        conv_bn(exp, p + ".conv1", p + ".bn1", planes, cin, 1); conv_bn(exp, p + ".conv2", p + ".bn2", planes, planes, 3)
        conv_bn(exp, p + ".conv3", p + ".bn3", planes * 4, planes, 1)
        if ds:
            conv_bn(exp, p + ".downsample.0", p + ".downsample.1", planes * 4, cin, 1)

    for k in range(4):
        bottleneck("layer1.%d" % k, 64 if k == 0 else 256, 64, k == 0)
    conv_bn(exp, "transition1.0.0", "transition1.0.1", 18, 256, 3); conv_bn(exp, "transition1.1.0.0", "transition1.1.0.1", 36, 256, 3)
    conv_bn(exp, "transition2.2.0.0", "transition2.2.0.1", 72, 36, 3); conv_bn(exp, "transition3.3.0.0", "transition3.3.0.1", 144, 72, 3)
    for stage, nmod, chans in (("stage2", 1, (18, 36)), ("stage3", 4, (18, 36, 72)), ("stage4", 3, (18, 36, 72, 144))):
        for m in range(nmod):
            p = "%s.%d" % (stage, m)
            for b, c in enumerate(chans):
                for k in range(4):
                    conv_bn(exp, "%s.branches.%d.%d.conv1" % (p, b, k), "%s.branches.%d.%d.bn1" % (p, b, k), c, c, 3)
                    conv_bn(exp, "%s.branches.%d.%d.conv2" % (p, b, k), "%s.branches.%d.%d.bn2" % (p, b, k), c, c, 3)
            for i, ci in enumerate(chans):
                for j, cj in enumerate(chans):
                    q = "%s.fuse_layers.%d.%d" % (p, i, j)
                    if j > i:
                        conv_bn(exp, q + ".0", q + ".1", ci, cj, 1)
                    elif j < i:
                        for k in range(i - j):
                            last = k == i - j - 1
                            conv_bn(exp, "%s.%d.0" % (q, k), "%s.%d.1" % (q, k), ci if last else cj, cj, 3)
    for i, (c, pl) in enumerate(zip((18, 36, 72, 144), (32, 64, 128, 256))):
        bottleneck("incre_modules.%d.0" % i, c, pl, True)
    got = {k[len("init_net.img_backbone."):]: list(v.shape) for k, v in sd.items() if ".img_backbone." in k}
    assert got == exp, (sorted(set(got) ^ set(exp))[:8])


def test_pc_normalize_matches_reference_definition():
    """aux_utils/pointnet2_utils.py:11-20 on the reference's own FPS keypoints (fixture copy of lmo obj_000001.pkl)"""
    import numpy as np
    from checkerpose_amd.aux_utils.pointnet2_utils import pc_normalize
    from tests.common import DATA, pc_normalize as ref_def
    xyz = np.load(os.path.join(DATA, "fps_lmo_obj01.npy"))[:512]
    out = pc_normalize(xyz.copy())
    assert np.array_equal(out, ref_def(xyz.copy()))
    assert np.abs(out.mean(0)).max() < 1e-12 and abs(np.sqrt((out ** 2).sum(1)).max() - 1.0) < 1e-12


def test_common_ops_helpers():
    from checkerpose_amd.common_ops import from_dim_str_to_tuple, get_batch_size
    assert get_batch_size(0.75, 32) == (8, 24)
    assert from_dim_str_to_tuple("1024_256_32") == (1024, 256, 32) and from_dim_str_to_tuple(None) is None


def _desc(**kw):
    d = _abi.CpConvDesc()
    base = dict(dtype=_abi.CP_F32, out_f32=0, B=1, H=8, W=16, Cin=16, in_cstride=16, in_coff=0, R=3, S=3, stride=1, pad=1,
                Ho=8, Wo=16, Cout=16, act=0, slope=0.0, o_base=0, o_sb=8 * 16 * 16, o_sy=16 * 16, o_sx=16, o_sc=1)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_descriptor_validation_without_gpu(lib):
    """Every conv entry point rejects bad descriptors BEFORE touching the device (fake, never-dereferenced pointers)."""
    A = 0x10000                        # 16-byte aligned fake device address
    ok = _desc()
    f = lib.cp_conv2d_igemm
    assert f(None, C.byref(_desc(dtype=7)), A, A, A, A, None, A) == -1                  # unknown dtype
    assert f(None, C.byref(_desc(Cin=18, in_cstride=18)), A, A, A, A, None, A) == -3    # channels not 16-byte multiples
    assert f(None, C.byref(_desc(Cout=18)), A, A, A, A, None, A) == -3                  # vector epilogue needs Cout % 4
    assert f(None, C.byref(_desc(in_coff=8, Cin=16, in_cstride=16)), A, A, A, A, None, A) == -3   # slice outside stride
    assert f(None, C.byref(ok), A + 4, A, A, A, None, A) == -3                          # misaligned input pointer
    assert f(None, C.byref(_desc(B=1 << 20, H=64, W=64, in_cstride=256, Cin=256)), A, A, A, A, None, A) == -4   # >= 2 GiB
    assert f(None, C.byref(_desc(stride=0)), A, A, A, A, None, A) == -1
    h = lib.cp_conv3x3_halo
    assert h(None, C.byref(_desc(stride=2)), A, A, A, A, None, A) == -1                 # halo kernel: 3x3/s1/p1 only
    assert h(None, C.byref(_desc(o_sc=2)), A, A, A, A, None, A) == -1
    assert h(None, C.byref(_desc(out_f32=1)), A, A, A, A, None, A) == -1
    g = lib.cp_gemm_rows
    assert g(None, C.byref(_desc(R=3, S=3)), A, A, A, A, None, A) == -1                 # gemm kernel: 1x1 only
    assert g(None, C.byref(_desc(R=1, S=1, pad=0, Cin=18, in_cstride=18)), A, A, A, A, None, A) == -3
    bb = lib.cp_basicblock_fused
    assert bb(None, C.byref(_desc(Cin=64, in_cstride=64, Cout=64)), A, A, A, A, A, A, A, A + 0x100000) == -1   # C > 32
    assert bb(None, C.byref(_desc()), A, A, A, A, A, A, A, A) == -1                     # in-place refused
    assert lib.cp_edgeconv_gather_max(None, 0, A, A, None, A, 1, 512, 20, 30, 1, 32, 0, 0.2) == -3      # C not a 16-B multiple
    assert lib.cp_edgeconv_gather_max(None, 0, A, A, None, A, 1, 512, 100, 64, 1, 64, 0, 0.2) == -1     # K > 64
    assert lib.cp_index2feat_gather(None, 0, A, A, A, A, A, 1, 512, 17, 17, 64, 2, 200, 0) == -3        # 4*E > out stride
    assert lib.cp_fuse_sum_act(None, 0, 5, None, None, A, 1, 8, 8, 16, 1, 16, 0) == -1                          # nsrc > 4
    assert lib.cp_upsample2x_bilinear_ac(None, 0, A, A, 1, 4, 4, 18, 20, 0, 20, 0) == -3
    assert lib.cp_maxpool3x3s2(None, 0, A, A, 1, 7, 8, 16) == -1                                          # odd height
    assert lib.cp_packed_halo_weight_bytes(_abi.CP_BF16, 256, 512) == 8 * 16 * 18 * 1024               # 8 groups x 16 chunks
    assert lib.cp_packed_halo_weight_bytes(_abi.CP_BF16, 18, 24) == 1 * 9 * 2 * 1024                    # small-Cout image
    assert lib.cp_packed_gemm_weight_bytes(_abi.CP_F32, 512, 256) == 16 * 16 * 2 * 1024


def test_training_entry_points_validate_before_launch(lib):
    """SURVEY 8f row N1 entry points: bad arguments are rejected before anything touches the device"""
    A = 0x10000
    wd = _abi.CpWgradDesc()
    for k, v in dict(dtype=_abi.CP_BF16, B=1, H=8, W=8, Ho=8, Wo=8, Cout=16, dy_cstride=16, dy_coff=0, Cin=16, x_cstride=16, x_coff=0,
                     R=3, S=3, stride=1, pad=1, dw_base=0, dw_sco=144, dw_sci=9, dw_sr=3, dw_ss=1).items():
        setattr(wd, k, v)
    assert lib.cp_conv2d_wgrad(None, C.byref(wd), None, A, A) == -1                                   # null dy
    wd.dy_cstride = 12
    assert lib.cp_conv2d_wgrad(None, C.byref(wd), A, A, A) == -3                                      # stride not a 16-byte multiple
    wd.dy_cstride, wd.Cout = 16, 24
    assert lib.cp_conv2d_wgrad(None, C.byref(wd), A, A, A) == -3                                      # channels outside the pixel row
    wd.Cout, wd.dtype = 16, 7
    assert lib.cp_conv2d_wgrad_ws(None, C.byref(wd), A, A, A, A, 1 << 20) == -1                       # unknown dtype
    assert lib.cp_weight_dgrad(None, None, 4, 4, 3, 3, A) == -1
    assert lib.cp_bn_train_stats(None, 0, A, 64, 18, 18, 0, None, None, None, None, 0.1, 1e-5, A, A, A, A, A) == -3   # cstride % 4
    assert lib.cp_bn_train_stats(None, 0, A, 0, 16, 16, 0, None, None, None, None, 0.1, 1e-5, A, A, A, A, A) == -1    # M == 0
    assert lib.cp_bn_train_bwd(None, 0, A, 16, 0, A, 16, 0, A, 16, 0, None, None, None, 64, 16, 1, 0.0, A, 16, 0, None, 0, 0, 0,
                               None, None, A) == -1                                                     # raw x without mean/rstd
    assert lib.cp_affine_act(None, 1, A, 16, 0, A, A, None, 0, 0, A + 2, 16, 0, 64, 16, 1, 0.0) == -3 # misaligned output
    assert lib.cp_edgeconv_train_fwd(None, 0, A, A, None, A, A, None, None, 0.1, 1e-5, A, 24, 0, A, A, A, A, A, A, 2, 64, 8, 24, 1, 0.2) == -3   # C % 16
    assert lib.cp_edgeconv_train_bwd(None, 0, A, A, None, None, None, A, 32, 0, A, A, 32, 0, A, A, A, A, A, A, A, 2, 64, 8, 32, 1, 0.2) == -1    # no reverse graph
    assert lib.cp_edge_weight_view(None, A, 8, 8, 2, A) == -1                                          # unknown mode
    assert lib.cp_upsample2x_bilinear_ac_bwd(None, 0, A, A, 1, 4, 4, 18, 20, 0, 20, 0, 0) == -3
    assert lib.cp_fuse_sum_act_bwd(None, 0, A, None, A, 1, 4, 4, 16, 1, 1, 0) == -1                    # relu mask without `out`
    assert lib.cp_maxpool3x3s2_bwd(None, 0, A, A, A, 1, 7, 8, 16, 0) == -1
    assert lib.cp_memset_zero(None, A + 4, 64) == -3 and lib.cp_memset_zero(None, None, 64) == -1
    assert lib.cp_memcpy_d2d(None, None, A, 16) == -1
    assert lib.cp_strided_to_nhwc(None, 0, A, 1, 0, 1, 1, 1, A, 1, 4, 3, 4) == -1                      # src dtype neither fp32 nor dtype
    assert lib.cp_index2feat_gather_bwd_t(None, 1, A, A, A, A, A, 1, 8, 9, 9, 6, 2, 24, 0) == -3       # E % 4
    assert lib.cp_bn_workspace_bytes(18) == 512 * 2 * 32 * 8 and lib.cp_bn_bwd_workspace_bytes(18) == 512 * 2 * 32 * 8 + 4 * 32 * 4
    assert lib.cp_edge_train_workspace_bytes(2, 64) == 512 * 2 * 64 * 8 + 4 * 64 * 4


def test_workspace_planner_never_aliases_live_tensors():
    """engine.Program.finalize(): linear-scan placement -- tensors whose live ranges overlap never overlap in memory,
    and fork/join regions pin every tensor they touch until the join (host logic only: no device needed)."""
    from checkerpose_amd.engine import Program, TBuf

    class _Lib:
        def cp_chan_align(self, dt):
            return 4

    class _WS:
        pass

    prog = Program.__new__(Program)
    prog.lib, prog.ws, prog.dtype, prog.B, prog.device = _Lib(), _WS(), 0, 2, "cpu"
    prog.E, prog.es, prog.ops, prog.tbufs, prog.keep = 4, 4, [], [], []
    prog.lane, prog.nlanes, prog.regions, prog._open, prog.flops, prog.conv_log = 0, 1, [], None, 0, []
    prog._rw = {}
    import random
    rnd = random.Random(0)
    live = []
    for i in range(60):
        if i == 20:
            prog.par_begin(3)
        if i == 40:
            prog.par_end()
        t = prog.tensor(rnd.randint(1, 50) * 64)
        reads = rnd.sample(live, min(len(live), rnd.randint(0, 3)))
        prog._add(lambda *a: 0, lambda P: (), "op%d" % i, reads, [t])
        live.append(t)
        if len(live) > 6:
            live.pop(rnd.randrange(len(live)))
    import torch
    orig_empty = torch.empty
    prog.finalize()
    placed = [t for t in prog.tbufs if t.first is not None]
    for a in placed:
        for b in placed:
            if a is not b and a.first <= b.last and b.first <= a.last:          # live ranges overlap
                assert a.offset + a.nbytes <= b.offset or b.offset + b.nbytes <= a.offset
    rs, re = prog.regions[0]
    assert all(t.last >= re for t in placed if t.first <= re and t.last >= rs)   # pinned until the join
    assert prog.workspace_bytes <= sum(t.nbytes for t in placed)


def test_dead_launch_elimination_keeps_exactly_what_is_read():
    """engine.Program.finalize(dce=True): a launch is kept iff it writes a caller-visible (fixed) tensor, declares no outputs,
    or feeds a kept launch; dropped launches take their conv_log entry and their workspace with them (host logic only)."""
    import torch
    from checkerpose_amd.engine import Program, _ConvLog

    class _Lib:
        def cp_chan_align(self, dt):
            return 4

    prog = Program.__new__(Program)
    prog.lib, prog.ws, prog.dtype, prog.B, prog.device = _Lib(), object(), 0, 2, "cpu"
    prog.E, prog.es, prog.ops, prog.tbufs, prog.keep = 4, 4, [], [], []
    prog.lane, prog.nlanes, prog.regions, prog._open, prog.flops = 0, 1, [], None, 0
    prog.conv_log, prog._rw = _ConvLog(prog), {}
    out = prog.fixed(torch.zeros(4))                       # caller-visible
    a, b, c, d, e = (prog.tensor(256) for _ in range(5))
    calls = []

    def add(name, reads, writes, flops=0):
        prog._add(lambda *x: 0, lambda P: (), name, reads, writes)
        if flops:
            prog.flops += flops
            prog.conv_log.append((name, 1, 1, 1, flops, "conv_igemm", 1))
        calls.append(name)

    add("in->a", [], [a], 10)
    add("a->b", [a], [b], 20)
    add("a->c (dead branch)", [a], [c], 40)
    add("c->d (dead branch)", [c], [d], 80)
    add("b->out", [b], [out], 160)
    add("side effect, no declared outputs", [b], [])
    add("writes e, never read", [], [e], 320)
    prog.finalize(dce=True)
    assert prog.dropped == 3
    assert [n for _, _, n in prog.calls] == ["in->a", "a->b", "b->out", "side effect, no declared outputs"]
    assert [r[0] for r in prog.conv_log] == ["in->a", "a->b", "b->out"] and prog.flops == 190
    assert c.first is None and d.first is None and e.first is None        # dead tensors get no workspace
    assert a.first == 0 and a.last == 1 and b.last == 5



def test_pretrained_backbone_is_never_silent(tmp_path, monkeypatch):
    """get_timm_backbone(pretrained=True) (reference backbone.py:48-49; pretrain.py:180-183 default): loads a local timm
    checkpoint when one is configured, otherwise warns (or raises on request) -- never a silent PyTorch-default init."""
    import warnings
    from checkerpose_amd.model.backbone import get_timm_backbone
    monkeypatch.delenv("CHECKERPOSE_AMD_TIMM_CKPT", raising=False)
    monkeypatch.delenv("CHECKERPOSE_AMD_TIMM_CKPT_DIR", raising=False)
    with pytest.warns(RuntimeWarning, match="no local timm checkpoint"):
        m = get_timm_backbone("resnet34", pretrained=True)
    assert float(m.layer1[0].bn2.weight.abs().max()) == 0.0          # timm's zero_init_last
    monkeypatch.setenv("CHECKERPOSE_AMD_REQUIRE_PRETRAINED", "1")
    with pytest.raises(RuntimeError, match="no local timm checkpoint"):
        get_timm_backbone("resnet34", pretrained=True)
    monkeypatch.delenv("CHECKERPOSE_AMD_REQUIRE_PRETRAINED")
    sd = {k: torch.full_like(v, 0.25) for k, v in m.state_dict().items()}
    sd["fc.weight"] = torch.zeros(1000, 512)                          # classifier keys of a timm checkpoint are dropped
    torch.save(sd, tmp_path / "resnet34.pth")
    monkeypatch.setenv("CHECKERPOSE_AMD_TIMM_CKPT_DIR", str(tmp_path))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m2 = get_timm_backbone("resnet34", pretrained=True)
    assert float(m2.conv1.weight.min()) == 0.25 and float(m2.layer4[2].bn2.running_var.max()) == 0.25


def test_neighbour_schedule_is_a_permutation_and_reduces_clashes():
    """graph_sched.schedule_neighbours: every keypoint keeps exactly its neighbours (the K-way max cannot change); the rows one
    step of cp_edgeconv_fused reads per 16-keypoint fragment share far fewer residues mod 16 than the kNN order does."""
    import numpy as np
    from checkerpose_amd.graph_sched import schedule_neighbours
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 512, size=(2, 512, 20)).astype(np.int32)
    out, before, after = schedule_neighbours(idx)
    assert out.shape == idx.shape and out.dtype == np.int32
    assert (np.sort(out, axis=2) == np.sort(idx, axis=2)).all()
    assert after * 2 < before
