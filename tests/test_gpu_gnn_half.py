"""GPU (-m gpu): the keypoint-side (GNN) kernels in IEEE half (CP_F16, round 6) through the C ABI, against torch with the SAME
roundings (inputs / weights / hidden rows to f16, fp32 accumulation) -- the half twins of the bf16 cases in test_gpu_parity.py.
Tolerances: f16 has 11 significant bits (bf16: 8), so the per-op bound is 4e-3 * (1 + |ref|) where the bf16 cases use 2e-2 .. 4e-2;
entry points that only move or re-type data are compared bit for bit.  End to end: the bf16 program with the keypoint side in half
must be CLOSER to the fp32 path than the all-bf16 program (that is what the mode is for).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from checkerpose_amd import _abi
from checkerpose_amd._abi import ACT_NONE, CP_BF16, CP_F16, CP_F32, CpConvDesc
from oracle import checkerpose_oracle as O
from tests.common import ape_p3d, build_net, det_image, det_tensor

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TDT = {CP_BF16: torch.bfloat16, CP_F16: torch.float16}
TOL_H = 4e-3


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def st():
    return torch.cuda.current_stream().cuda_stream


def rnd(x, dtype):
    return x.to(TDT[dtype]).float()


def close(a, b, tol):
    a, b = a.double(), b.double()
    err = ((a - b).abs() / (1 + b.abs())).max().item()
    assert err <= tol, "max rel err %.3e > %.1e" % (err, tol)


def _gemm_pack(lib, dtype, w, co, ci):
    buf = torch.empty(lib.cp_packed_gemm_weight_bytes(dtype, co, ci), dtype=torch.uint8, device=dev())
    wd = w.contiguous().to(dev())
    _abi.check(lib.cp_pack_gemm_weight(st(), dtype, wd.data_ptr(), co, ci, ci, buf.data_ptr()))
    torch.cuda.synchronize()
    return buf


@pytest.mark.parametrize("Cc", [64, 256])
def test_edgeconv_fused_half_vs_torch(lib, Cc):
    """cp_edgeconv_fused_t(CP_F16): out = leaky(max_k f16(s W1 x_j(k)) + s (W2 - W1) x_i + t) with x, W in f16 and fp32 accumulation,
    on the real `ape` graph, mixed-sign folded scales, a channel-sliced output"""
    B, N, K = 3, 512, 20
    idx = O.knn(ape_p3d(512), K)
    x = rnd(det_tensor("hx%d" % Cc, (B, N, Cc)), CP_F16)
    w1 = rnd(det_tensor("hw1%d" % Cc, (Cc, Cc), (3.0 / Cc) ** 0.5), CP_F16)
    dq = rnd(det_tensor("hdq%d" % Cc, (Cc, Cc), (3.0 / Cc) ** 0.5), CP_F16)
    s_ = 1.0 + 0.3 * det_tensor("hs%d" % Cc, (Cc,))
    s_[::5] *= -1.0
    t_ = det_tensor("ht%d" % Cc, (Cc,), 0.5)
    P = ((x @ w1.t()) * s_).to(torch.float16).float()
    Q = (x @ dq.t()) * s_ + t_
    ref = F.leaky_relu(P[:, idx[0]].max(dim=2)[0] + Q, 0.2)
    wpq = torch.cat([w1, dq], 0).contiguous().to(dev())
    scale, shift = torch.cat([s_, s_]).to(dev()), torch.cat([torch.zeros_like(t_), t_]).to(dev())
    pk = torch.empty(lib.cp_edgeconv_fused_weight_bytes(Cc, Cc), dtype=torch.uint8, device=dev())
    _abi.check(lib.cp_pack_edgeconv_fused_weight_t(st(), CP_F16, wpq.data_ptr(), Cc, Cc, pk.data_ptr()))
    xd = x.to(torch.float16).to(dev())
    idx_d = idx.to(torch.int32).contiguous().to(dev())
    wide = torch.full((B, N, Cc + 64), 7.0, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_edgeconv_fused_t(st(), CP_F16, xd.data_ptr(), Cc, 0, pk.data_ptr(), scale.data_ptr(), shift.data_ptr(), idx_d.data_ptr(),
                                       None, wide.data_ptr(), Cc + 64, 64, B, N, K, Cc, Cc, 1, 0.2))
    torch.cuda.synchronize()
    got = wide.float().cpu()
    assert float((got[..., :64] - 7.0).abs().max()) == 0.0
    close(got[..., 64:], ref, TOL_H)
    # the bf16 twin of the same call keeps working and is the coarser of the two
    assert lib.cp_edgeconv_fused_t(st(), 7, xd.data_ptr(), Cc, 0, pk.data_ptr(), scale.data_ptr(), shift.data_ptr(), idx_d.data_ptr(), None,
                                   wide.data_ptr(), Cc + 64, 64, B, N, K, Cc, Cc, 1, 0.2) == -1          # unknown dtype: refused


@pytest.mark.parametrize("B,N,Cin", [(3, 512, 512), (2, 100, 320), (2, 200, 128)])
def test_mlp_pair_fused_half_vs_torch(lib, B, N, Cin):
    """cp_mlp_pair_fused_t(CP_F16) == two Linear + LeakyReLU in torch with f16 inputs, weights, hidden rows and output"""
    w = [det_tensor("hp_w1_%d" % Cin, (256, Cin), (6.0 / Cin) ** 0.5), det_tensor("hp_w2", (256, 256), (6.0 / 256) ** 0.5)]
    b = [det_tensor("hp_b1", (256,), 0.1), det_tensor("hp_b2", (256,), 0.1)]
    x = rnd(det_tensor("hp_x%d_%d" % (Cin, N), (B, N, Cin)), CP_F16)
    h1 = rnd(F.leaky_relu(x @ rnd(w[0], CP_F16).t() + b[0], 0.01), CP_F16)
    ref = F.leaky_relu(h1 @ rnd(w[1], CP_F16).t() + b[1], 0.01)
    xin = x.to(torch.float16).to(dev())
    pk = [_gemm_pack(lib, CP_F16, w[0], 256, Cin), _gemm_pack(lib, CP_F16, w[1], 256, 256)]
    bd = [t.contiguous().to(dev()) for t in b]
    wide = torch.full((B, N, 320), 7.0, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_mlp_pair_fused_t(st(), CP_F16, xin.data_ptr(), Cin, 0, Cin, B, N, pk[0].data_ptr(), bd[0].data_ptr(), 0.01,
                                       pk[1].data_ptr(), bd[1].data_ptr(), 0.01, wide.data_ptr(), 320, 64))
    torch.cuda.synchronize()
    got = wide.float().cpu()
    assert float((got[..., :64] - 7.0).abs().max()) == 0.0
    close(got[..., 64:], ref, TOL_H)


@pytest.mark.parametrize("Cg,H", [(64, 16), (256, 32)])
def test_mlp_pair_fused_gather_half_bitwise_equals_gather_plus_pair(lib, Cg, H):
    """cp_mlp_pair_fused_gather_t(CP_F16): Index2Feat's 4-tap gather x RoI bit done by the loader from a HALF patch map == the half
    pair kernel on the materialised [taps | graph] rows, bit for bit (the loader only moves bytes); border ids, a masked-out row"""
    B, N, E, k = 2, 512, 64, 2
    Hp = H + 1
    Cin = 256 + Cg
    patches = det_tensor("hg_p%d" % H, (B, Hp, Hp, E)).to(torch.float16)
    graph = det_tensor("hg_g%d" % Cg, (B, N, Cg)).to(torch.float16)
    g_ = torch.Generator().manual_seed(5)
    xid = torch.randint(0, H // 2, (B, N), generator=g_, dtype=torch.int32)
    yid = torch.randint(0, H // 2, (B, N), generator=g_, dtype=torch.int32)
    xid[0, :4] = torch.tensor([0, H // 2 - 1, 0, H // 2 - 1]); yid[0, :4] = torch.tensor([0, 0, H // 2 - 1, H // 2 - 1])
    mask = torch.ones(B, N); mask[1, 7] = 0.0
    taps = []
    for t in range(4):
        yy = (2 * yid.long() + (k if (t & 1) else 0)).clamp(max=Hp - 1)
        xx = (2 * xid.long() + (k if (t & 2) else 0)).clamp(max=Hp - 1)
        taps.append(patches[torch.arange(B)[:, None], yy, xx])                # (B, N, E)
    local = torch.cat(taps, -1) * mask[..., None].to(torch.float16)
    cat = torch.cat([local, graph], -1).contiguous().to(dev())             # (B, N, Cin)
    w = [det_tensor("hg_w1_%d" % Cin, (256, Cin), (6.0 / Cin) ** 0.5), det_tensor("hg_w2", (256, 256), (6.0 / 256) ** 0.5)]
    b = [det_tensor("hg_b1", (256,), 0.1), det_tensor("hg_b2", (256,), 0.1)]
    pk = [_gemm_pack(lib, CP_F16, w[0], 256, Cin), _gemm_pack(lib, CP_F16, w[1], 256, 256)]
    bd = [t.contiguous().to(dev()) for t in b]
    out_a = torch.zeros(B, N, 256, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_mlp_pair_fused_t(st(), CP_F16, cat.data_ptr(), Cin, 0, Cin, B, N, pk[0].data_ptr(), bd[0].data_ptr(), 0.01,
                                       pk[1].data_ptr(), bd[1].data_ptr(), 0.01, out_a.data_ptr(), 256, 0))
    g = _abi.CpI2fGather()
    pd, xd, yd, md = patches.contiguous().to(dev()), xid.to(dev()), yid.to(dev()), mask.to(dev())
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev())
    cat2 = torch.zeros(B, N, Cin, dtype=torch.float16, device=dev())
    cat2[..., 256:] = graph.to(dev())
    g.patches, g.x_id, g.y_id, g.mask, g.zeros = pd.data_ptr(), xd.data_ptr(), yd.data_ptr(), md.data_ptr(), zeros.data_ptr()
    g.p_cstride, g.p_coff, g.Hp, g.Wp, g.k = E, 0, Hp, Hp, k
    out_b = torch.zeros(B, N, 256, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_mlp_pair_fused_gather_t(st(), CP_F16, C.byref(g), cat2.data_ptr(), Cin, 256, Cg, B, N, pk[0].data_ptr(), bd[0].data_ptr(),
                                              0.01, pk[1].data_ptr(), bd[1].data_ptr(), 0.01, out_b.data_ptr(), 256, 0))
    torch.cuda.synchronize()
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16))
    x = cat.float().cpu()
    h1 = rnd(F.leaky_relu(x @ rnd(w[0], CP_F16).t() + b[0], 0.01), CP_F16)
    close(out_b.float().cpu(), F.leaky_relu(h1 @ rnd(w[1], CP_F16).t() + b[1], 0.01), TOL_H)


@pytest.mark.parametrize("B,N", [(3, 512), (2, 100)])
def test_mlp_query_fused_half_vs_torch(lib, B, N):
    """cp_mlp_query_fused_t(CP_F16) == the three Linear layers with f16 inputs / weights / first hidden rows, fp32 from there on"""
    net = build_net(seed=0)
    sd = net.state_dict()
    pfx = "refine_net.1.query_block.mlps."
    w = [sd[pfx + "%d.weight" % j].float() for j in (0, 2, 4)]
    b = [sd[pfx + "%d.bias" % j].float() for j in (0, 2, 4)]
    x = rnd(det_tensor("hq%d_%d" % (B, N), (B, N, 256)), CP_F16)
    h1 = rnd(F.leaky_relu(x @ rnd(w[0], CP_F16).t() + b[0], 0.01), CP_F16)
    h2 = F.leaky_relu(h1 @ rnd(w[1], CP_F16).t() + b[1], 0.01)
    ref = h2 @ w[2].t() + b[2]
    wide = torch.zeros(B, N, 320, dtype=torch.float16, device=dev())
    wide[..., 64:] = x.to(torch.float16).to(dev())
    pk = [_gemm_pack(lib, CP_F16, w[0], 256, 256), _gemm_pack(lib, CP_F16, w[1], 64, 256)]
    ones = [torch.ones(256, device=dev()), torch.ones(64, device=dev())]
    bd = [t.contiguous().to(dev()) for t in b]
    w3 = w[2].contiguous().to(dev())
    bits = torch.full((B, 13, N), 7.0, device=dev())
    _abi.check(lib.cp_mlp_query_fused_t(st(), CP_F16, wide.data_ptr(), 320, 64, B, N, pk[0].data_ptr(), ones[0].data_ptr(), bd[0].data_ptr(), 0.01,
                                        pk[1].data_ptr(), ones[1].data_ptr(), bd[1].data_ptr(), 0.01, w3.data_ptr(), bd[2].data_ptr(),
                                        bits.data_ptr(), 5 * N, 13 * N, 1, 6 * N))
    torch.cuda.synchronize()
    got = bits.cpu()
    keep = [r for r in range(13) if r not in (5, 11)]
    assert float((got[:, keep] - 7.0).abs().max()) == 0.0
    close(torch.stack([got[:, 5], got[:, 11]], -1), ref, 5e-4)


def test_index2feat_conv_half_output_is_the_same_sum_rounded_to_half(lib):
    """cp_index2feat_conv_t(out_dtype): the conv itself stays bf16 x bf16 -> fp32; only the output rounding changes.  The half rows
    must be the finer rounding of the same fp32 sums: |half - bf16| <= one bf16 ulp of the value, and half == torch's conv (bf16
    inputs, fp32 sums) to f16 precision"""
    net = build_net(seed=0)
    sd = net.state_dict()
    B, N, H = 2, 512, 64
    f = det_tensor("hi2f", (B, 256, H, H))
    w = sd["refine_net.2.local_feat_ext_block.patch_generator.weight"]
    bias = sd["refine_net.2.local_feat_ext_block.patch_generator.bias"]
    d = dev()
    pw = torch.empty(lib.cp_index2feat_conv_weight_bytes(), dtype=torch.uint8, device=d)
    wd = w.contiguous().to(d)
    _abi.check(lib.cp_pack_index2feat_conv_weight(st(), wd.data_ptr(), pw.data_ptr()))
    fin = f.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(d)
    g_ = torch.Generator().manual_seed(3)
    xid = torch.randint(0, 32, (B, N), generator=g_, dtype=torch.int32).to(d)
    yid = torch.randint(0, 32, (B, N), generator=g_, dtype=torch.int32).to(d)
    mask = torch.ones(B, N, device=d); mask[1, 8] = 0.0
    bd = bias.contiguous().to(d)
    outs = {}
    for dt in (CP_BF16, CP_F16):
        out = torch.full((B, N, 320), 5.0, dtype=TDT[dt], device=d)
        _abi.check(lib.cp_index2feat_conv_t(st(), dt, fin.data_ptr(), 256, 0, pw.data_ptr(), bd.data_ptr(), xid.data_ptr(), yid.data_ptr(),
                                            mask.data_ptr(), out.data_ptr(), B, N, H, H, 2, 320, 64), "index2feat conv")
        torch.cuda.synchronize()
        assert float((out[..., :64].float() - 5.0).abs().max()) == 0.0
        outs[dt] = out[..., 64:].float().cpu()
    conv = F.conv2d(f.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), bias, 1, 1)        # (B, 64, H + 1, H + 1)
    xi, yi = xid.cpu().long(), yid.cpu().long()
    taps = [conv[torch.arange(B)[:, None], :, 2 * yi + (2 if (t & 1) else 0), 2 * xi + (2 if (t & 2) else 0)] for t in range(4)]
    ref = torch.cat(taps, -1) * mask.cpu()[..., None]
    close(outs[CP_F16], ref, TOL_H)
    assert float(((outs[CP_F16] - outs[CP_BF16]).abs() / (ref.abs() + 1e-3)).max()) <= 2.0 ** -7
    assert float(outs[CP_F16][1, 8].abs().max()) == 0.0


def _desc(dtype, out_f32, B, H, W, cin, cout, R, pad, Ho, Wo):
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = dtype, out_f32, B, H, W
    d.Cin, d.in_cstride, d.in_coff = cin, cin, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = R, R, 1, pad, Ho, Wo, cout, ACT_NONE, 0.0
    d.ksplit = -1
    return d


def test_conv2x2_halo_half_output(lib):
    """cp_conv2x2_halo with CpConvDesc.out_f32 = 2: the bf16 patch_generator conv (pipeline.py:144-145) writing IEEE-half rows"""
    B, Cin, H, W, Cout = 2, 256, 16, 16, 64
    x = det_tensor("h2x", (B, Cin, H, W))
    w = det_tensor("h2w", (Cout, Cin, 2, 2), (2.0 / (Cin * 4)) ** 0.5 * 1.7)
    shift = 0.2 * det_tensor("h2t", (Cout,))
    ref = F.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), None, 1, 1) + shift.view(1, -1, 1, 1)
    xin = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(dev())
    pw = torch.empty(lib.cp_packed_conv2x2_halo_weight_bytes(CP_BF16, Cout, Cin), dtype=torch.uint8, device=dev())
    wd = w.contiguous().to(dev())
    _abi.check(lib.cp_pack_conv2x2_halo_weight(st(), CP_BF16, wd.data_ptr(), Cout, Cin, Cin, pw.data_ptr()))
    sc, sh = torch.ones(Cout, device=dev()), shift.to(dev())
    Ho, Wo = H + 1, W + 1
    d = _desc(CP_BF16, 2, B, H, W, Cin, Cout, 2, 1, Ho, Wo)
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, Ho * Wo * Cout, Wo * Cout, Cout, 1
    out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float16, device=dev())
    _abi.check(lib.cp_conv2x2_halo(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()))
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    close(out.float().cpu().permute(0, 3, 1, 2), ref, TOL_H)
    d.dtype = CP_F32                                                      # half rows are an option of the bf16 conv only
    assert lib.cp_conv2x2_halo(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()) == -1


def test_conv2d_igemm_half_edges(lib):
    """cp_conv2d_igemm at the keypoint side's two generic-kernel edges: (1) conv1x1 1024 -> N as a bf16 conv whose strided epilogue
    writes HALF rows in the (B, N, 64) graph layout (init.py:112-114); (2) Linear(64 -> 7) reading HALF rows with half weights, fp32
    logits out (init.py:120-122)"""
    B, Cb, N = 2, 1024, 512
    f = det_tensor("hc1f", (B, Cb, 8, 8))
    w = det_tensor("hc1w", (N, Cb, 1, 1), (1.0 / Cb) ** 0.5)
    bias = det_tensor("hc1b", (N,), 0.1)
    ref = (F.conv2d(f.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), bias)).view(B, N, 64)
    xin = f.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(dev())
    pw = torch.empty(lib.cp_packed_weight_bytes(CP_BF16, N, Cb, 1, 1), dtype=torch.uint8, device=dev())
    wd = w.contiguous().to(dev())
    _abi.check(lib.cp_pack_conv_weight(st(), CP_BF16, wd.data_ptr(), N, Cb, 1, 1, Cb, 0, 0, None, N, pw.data_ptr()))
    sc, sh = torch.ones(N, device=dev()), bias.to(dev())
    d = _desc(CP_BF16, 2, B, 8, 8, Cb, N, 1, 0, 8, 8)
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, N * 64, 8, 1, 64
    g0 = torch.full((B, N, 64), float("nan"), dtype=torch.float16, device=dev())
    _abi.check(lib.cp_conv2d_igemm(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, g0.data_ptr()))
    torch.cuda.synchronize()
    close(g0.float().cpu(), ref, TOL_H)
    # (2)
    g = rnd(det_tensor("hl_g", (B, N, 64)), CP_F16)
    wl = det_tensor("hl_w", (7, 64), 0.125)
    bl = det_tensor("hl_b", (7,), 0.1)
    ref2 = (g @ rnd(wl, CP_F16).t() + bl).permute(0, 2, 1)                  # (B, 7, N)
    gd = g.to(torch.float16).contiguous().to(dev())
    pw2 = torch.empty(lib.cp_packed_weight_bytes(CP_F16, 16, 64, 1, 1), dtype=torch.uint8, device=dev())
    wd2 = wl.contiguous().to(dev())
    _abi.check(lib.cp_pack_conv_weight(st(), CP_F16, wd2.data_ptr(), 7, 64, 1, 1, 64, 0, 0, None, 7, pw2.data_ptr()))
    sc2 = torch.zeros(16, device=dev()); sc2[:7] = 1.0
    sh2 = torch.zeros(16, device=dev()); sh2[:7] = bl.to(dev())
    d2 = _desc(CP_F16, 1, B, 1, N, 64, 7, 1, 0, 1, N)
    d2.o_base, d2.o_sb, d2.o_sy, d2.o_sx, d2.o_sc = 0, 7 * N, 0, 1, N
    bits = torch.full((B, 7, N), float("nan"), device=dev())
    _abi.check(lib.cp_conv2d_igemm(st(), C.byref(d2), gd.data_ptr(), pw2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), None, bits.data_ptr()))
    torch.cuda.synchronize()
    close(bits.cpu(), ref2, 2e-4)
    res = torch.zeros(B, 7, N, device=dev())                               # a residual is not part of the half edges: refused
    assert lib.cp_conv2d_igemm(st(), C.byref(d2), gd.data_ptr(), pw2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), res.data_ptr(), bits.data_ptr()) == -1


def test_bf16_program_with_the_keypoint_side_in_half_is_closer_to_fp32(monkeypatch):
    """The mode's purpose, end to end: at the bench's kernel selection the bf16 program whose keypoint side runs in IEEE half
    (engine.USE_GNN_F16, the default) has a smaller teacher-forced logit error against the fp32 path than the all-bf16 program --
    and its launch list really carries the half kernels (no silent bf16)."""
    from checkerpose_amd import engine
    B = 4
    img = det_image(B, seed=2).cuda()
    net = build_net(seed=1).cuda().eval()
    net.set_kernel_selection("per_crop")
    net.set_compute_dtype("fp32")
    ref = [t.clone() for t in net(img, None)]
    t = torch.zeros(B, 13, 512, device=img.device)
    t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
    err = {}
    for half in (True, False):
        monkeypatch.setattr(engine, "USE_GNN_F16", half)
        net.set_compute_dtype("bf16")
        out = net.forward_teacher_forced(img, t)
        z, zr = torch.cat(out[:3], 1), torch.cat(ref[:3], 1)
        err[half] = float((z - zr).abs().mean())
        pr = [p for k, p in net._programs.items() if k[3] == "bf16"][-1]["prog"].progs[0]
        assert pr.gnn_half is half
    print("teacher-forced mean |dlogit| vs fp32: keypoint side half %.5f, bf16 %.5f" % (err[True], err[False]))
    assert err[True] < 0.9 * err[False], err


@pytest.mark.parametrize("N,Cc", [(1024, 256), (4096, 64)])
def test_edgeconv_tiled_half_vs_torch(lib, N, Cc):
    """cp_edgeconv_tiled_t(CP_F16) (N > 512: key-table launch + per-patch gather launch) with x / W in half == the factored form in
    torch with the same roundings, on real LM graphs in the patch order of graph_sched.tile_schedule"""
    import numpy as np
    from checkerpose_amd.graph_sched import tile_schedule
    from tests.common import lm_p3d
    objs = [0, 4]
    P3 = lm_p3d(N)[objs]
    idx = O.knn(P3, 20)                                                  # (2, N, K) original numbering
    sc = tile_schedule(idx.numpy(), P3.numpy())
    assert sc is not None and lib.cp_edgeconv_tiled_supported(N, 20, Cc, Cc, sc["HPAD"])
    gsel = torch.tensor([1, 0, 1])
    B, K = 3, 20
    x = rnd(det_tensor("htx%d_%d" % (Cc, N), (B, N, Cc)), CP_F16)
    w1 = rnd(det_tensor("htw1%d" % Cc, (Cc, Cc), (3.0 / Cc) ** 0.5), CP_F16)
    dq = rnd(det_tensor("htdq%d" % Cc, (Cc, Cc), (3.0 / Cc) ** 0.5), CP_F16)
    s_ = 1.0 + 0.3 * det_tensor("hts%d" % Cc, (Cc,))
    s_[::7] *= -1.0
    t_ = det_tensor("htt%d" % Cc, (Cc,), 0.5)
    Pk = ((x @ w1.t()) * s_).to(torch.float16).float()
    Q = (x @ dq.t()) * s_ + t_
    ref = torch.stack([F.leaky_relu(Pk[i][idx[gsel[i]]].max(dim=1)[0] + Q[i], 0.2) for i in range(B)])      # (B, N, C) original order
    wpq = torch.cat([w1, dq], 0).contiguous().to(dev())
    scale, shift = torch.cat([s_, s_]).to(dev()), torch.cat([torch.zeros_like(t_), t_]).to(dev())
    pf = torch.empty(lib.cp_edgeconv_fused_weight_bytes(Cc, Cc), dtype=torch.uint8, device=dev())
    pq = torch.empty(lib.cp_edgeconv_tiled_weight_bytes(Cc, Cc), dtype=torch.uint8, device=dev())
    _abi.check(lib.cp_pack_edgeconv_fused_weight_t(st(), CP_F16, wpq.data_ptr(), Cc, Cc, pf.data_ptr()))
    _abi.check(lib.cp_pack_edgeconv_tiled_weight_t(st(), CP_F16, wpq.data_ptr(), Cc, Cc, pq.data_ptr()))
    perm = torch.from_numpy(sc["perm"]).to(dev())
    halo, nbr = torch.from_numpy(sc["halo"]).contiguous().to(dev()), torch.from_numpy(sc["nbr"]).contiguous().to(dev())
    gids = gsel.to(torch.int32).to(dev())
    x_orig = x.to(torch.float16).contiguous().to(dev())
    x_int = torch.empty_like(x_orig)
    _abi.check(lib.cp_permute_rows(st(), x_orig.data_ptr(), x_int.data_ptr(), perm.data_ptr(), gids.data_ptr(), B, N, Cc * 2))
    ktab = torch.empty(lib.cp_edgeconv_tiled_table_bytes(B, N, Cc), dtype=torch.uint8, device=dev())
    out = torch.full((B, N, Cc), 7.0, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_edgeconv_tiled_t(st(), CP_F16, x_int.data_ptr(), Cc, 0, pf.data_ptr(), pq.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                       halo.data_ptr(), nbr.data_ptr(), gids.data_ptr(), ktab.data_ptr(), out.data_ptr(), Cc, 0,
                                       B, N, K, Cc, Cc, 2, int(sc["HPAD"]), 0.2))
    torch.cuda.synchronize()
    got_int = out.float().cpu()
    got = torch.empty_like(got_int)
    for i in range(B):
        got[i][sc["perm"][gsel[i]].astype(np.int64)] = got_int[i]        # internal row r holds original keypoint perm[r]
    close(got, ref, TOL_H)


# ---- the generic kernels in half: what the small-batch programs (below the per-crop crossovers) run the keypoint side on
@pytest.mark.parametrize("B,N,Cin,Cout,act", [(2, 512, 256, 512, 0), (64, 512, 256, 512, 0), (3, 200, 64, 128, 2), (1, 512, 320, 256, 2)])
def test_gemm_rows_half_vs_torch(lib, B, N, Cin, Cout, act):
    """cp_gemm_rows with CpConvDesc.dtype = CP_F16 (rows, weights and output in half; the weight-stationary variant from 16 384 rows
    on): the EdgeConv node GEMM 256 -> 512 and the per-keypoint Linears of a small-batch program"""
    x = rnd(det_tensor("hgx%d_%d_%d" % (B, N, Cin), (B, N, Cin)), CP_F16)
    w = det_tensor("hgw%d_%d" % (Cin, Cout), (Cout, Cin), (2.0 / Cin) ** 0.5 * 1.7)
    scale = 1.0 + 0.3 * det_tensor("hgs%d" % Cout, (Cout,))
    shift = 0.2 * det_tensor("hgt%d" % Cout, (Cout,))
    ref = (x @ rnd(w, CP_F16).t()) * scale + shift
    ref = F.leaky_relu(ref, 0.01) if act == 2 else ref
    xin = x.to(torch.float16).contiguous().to(dev())
    pw = _gemm_pack(lib, CP_F16, w, Cout, Cin)
    sc, sh = scale.to(dev()), shift.to(dev())
    d = _desc(CP_F16, 0, B, 1, N, Cin, Cout, 1, 0, 1, N)
    d.act, d.slope = act, 0.01
    d.ksplit = 0
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, N * Cout, N * Cout, Cout, 1
    out = torch.full((B, N, Cout), float("nan"), dtype=torch.float16, device=dev())
    _abi.check(lib.cp_gemm_rows(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()))
    torch.cuda.synchronize()
    close(out.float().cpu(), ref, TOL_H)
    res = torch.zeros_like(out)
    assert lib.cp_gemm_rows(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), out.data_ptr()) == -1


@pytest.mark.parametrize("B,N,Cin,Cout,ksplit", [(1, 512, 256, 512, 0), (1, 512, 256, 512, -1), (40, 512, 64, 128, 0), (2, 100, 256, 64, 0)])
def test_conv_igemm_half_rows_tiled_and_splitk(lib, B, N, Cin, Cout, ksplit):
    """cp_conv2d_igemm with dtype = CP_F16 and out_f32 = 0: half rows in, half weights, half rows out -- the tiled kernel (ksplit -1 /
    large M) and the split-K variant the small-batch programs take (ksplit 0, small M)"""
    x = rnd(det_tensor("hix%d_%d_%d" % (B, N, Cin), (B, N, Cin)), CP_F16)
    w = det_tensor("hiw%d_%d" % (Cin, Cout), (Cout, Cin), (2.0 / Cin) ** 0.5 * 1.7)
    shift = 0.2 * det_tensor("hit%d" % Cout, (Cout,))
    ref = F.leaky_relu(x @ rnd(w, CP_F16).t() + shift, 0.2)
    xin = x.to(torch.float16).contiguous().to(dev())
    pw = torch.empty(lib.cp_packed_weight_bytes(CP_F16, Cout, Cin, 1, 1), dtype=torch.uint8, device=dev())
    wd = w.contiguous().to(dev())
    _abi.check(lib.cp_pack_conv_weight(st(), CP_F16, wd.data_ptr(), Cout, Cin, 1, 1, Cin, 0, 0, None, Cout, pw.data_ptr()))
    sc, sh = torch.ones(Cout, device=dev()), shift.to(dev())
    d = _desc(CP_F16, 0, B, 1, N, Cin, Cout, 1, 0, 1, N)
    d.act, d.slope, d.ksplit = 2, 0.2, ksplit
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, N * Cout, N * Cout, Cout, 1
    out = torch.full((B, N, Cout), float("nan"), dtype=torch.float16, device=dev())
    _abi.check(lib.cp_conv2d_igemm(st(), C.byref(d), xin.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()))
    torch.cuda.synchronize()
    sym = lib.cp_last_kernel().decode()
    assert "F16Tag" in sym and (("splitk" in sym) == bool(ksplit == 0 and lib.cp_conv2d_igemm_splitk(CP_F16, B * N, Cin, Cout))), sym
    close(out.float().cpu(), ref, TOL_H)


@pytest.mark.parametrize("Cc", [64, 256])
def test_edgeconv_gather_max_half_vs_torch(lib, Cc):
    """cp_edgeconv_gather_max(CP_F16): out = leaky(max_k P'[j(k)] + Q'[i]) over a half [P' | Q'] table (the L2 gather of the small-batch
    programs), real `ape` graph, channel-sliced output"""
    B, N, K = 3, 512, 20
    idx = O.knn(ape_p3d(512), K)
    pq = rnd(det_tensor("hgm%d" % Cc, (B, N, 2 * Cc)), CP_F16)
    ref = F.leaky_relu(pq[:, idx[0], :Cc].max(dim=2)[0] + pq[..., Cc:], 0.2)
    pd = pq.to(torch.float16).contiguous().to(dev())
    idx_d = idx.to(torch.int32).contiguous().to(dev())
    wide = torch.full((B, N, Cc + 64), 7.0, dtype=torch.float16, device=dev())
    _abi.check(lib.cp_edgeconv_gather_max(st(), CP_F16, pd.data_ptr(), idx_d.data_ptr(), None, wide.data_ptr(), B, N, K, Cc, 1, Cc + 64, 64, 0.2))
    torch.cuda.synchronize()
    got = wide.float().cpu()
    assert float((got[..., :64] - 7.0).abs().max()) == 0.0
    close(got[..., 64:], ref, 1e-3)                                       # one half rounding of the sum


def test_index2feat_gather_half_is_a_byte_exact_gather(lib):
    """cp_index2feat_gather(CP_F16): taps x {0, 1} RoI bit of a half patch map == the torch gather, bit for bit"""
    B, N, E, k, H = 2, 512, 64, 2, 32
    Hp = H + 1
    patches = det_tensor("hig_p", (B, Hp, Hp, E)).to(torch.float16)
    g_ = torch.Generator().manual_seed(9)
    xid = torch.randint(0, H // 2, (B, N), generator=g_, dtype=torch.int32)
    yid = torch.randint(0, H // 2, (B, N), generator=g_, dtype=torch.int32)
    mask = (torch.rand(B, N, generator=g_) > 0.3).float()
    taps = []
    for t in range(4):
        yy = 2 * yid.long() + (k if (t & 1) else 0)
        xx = 2 * xid.long() + (k if (t & 2) else 0)
        taps.append(patches[torch.arange(B)[:, None], yy, xx])
    ref = torch.cat(taps, -1) * mask[..., None].to(torch.float16)
    out = torch.full((B, N, 320), 5.0, dtype=torch.float16, device=dev())
    pd, xd, yd, md = patches.contiguous().to(dev()), xid.to(dev()), yid.to(dev()), mask.to(dev())
    _abi.check(lib.cp_index2feat_gather(st(), CP_F16, pd.data_ptr(), xd.data_ptr(), yd.data_ptr(), md.data_ptr(), out.data_ptr(), B, N, Hp, Hp, E, k, 320, 64))
    torch.cuda.synchronize()
    assert torch.equal(out[..., 64:].cpu().view(torch.int16), ref.view(torch.int16).where(ref != 0, torch.zeros((), dtype=torch.int16)))
    assert float((out[..., :64].float() - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("selection", ["tiled", "auto"])
def test_small_batch_program_runs_the_keypoint_side_in_half_too(monkeypatch, selection):
    """the generic-kernel programs (small batches; `set_kernel_selection("tiled")`) run the keypoint side in half as well: closer to the
    fp32 path than with the flag off, and the launch list shows half kernels on the keypoint side"""
    from checkerpose_amd import engine
    B = 2
    img = det_image(B, seed=4).cuda()
    net = build_net(seed=1).cuda().eval()
    net.set_kernel_selection(selection)
    net.set_compute_dtype("fp32")
    ref = [t.clone() for t in net(img, None)]
    t = torch.zeros(B, 13, 512, device=img.device)
    t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
    err = {}
    for half in (True, False):
        monkeypatch.setattr(engine, "USE_GNN_F16", half)
        net.set_compute_dtype("bf16")
        out = net.forward_teacher_forced(img, t)
        z, zr = torch.cat(out[:3], 1), torch.cat(ref[:3], 1)
        err[half] = float((z - zr).abs().mean())
        pr = [p for k, p in net._programs.items() if k[3] == "bf16"][-1]["prog"].progs[0]
        assert pr.gnn_half is half
        names = [c[2].split(":")[0] for c in pr.calls]
        assert "edge_gather" in names and "edge_fused" not in names          # the small-batch EdgeConv path
    print("%s selection, B = %d: teacher-forced mean |dlogit| vs fp32: half %.5f, bf16 %.5f" % (selection, B, err[True], err[False]))
    assert err[True] < 0.9 * err[False], err
