"""GPU parity of the training-side ops through the C ABI: HIP vs the reference-produced
golden vectors and vs oracle/train_oracle.py on the same seeded inputs; size-independent properties at full size."""
import os

import numpy as np
import pytest
import torch

from tests import train_cases as TC
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_ops.npz"))
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _grad_on():
    """other test modules switch autograd off globally (inference parity); these tests need it"""
    with torch.enable_grad():
        yield


@pytest.mark.parametrize("name", list(TC.CODE_CASES))
def test_code_loss(name):
    from checkerpose_amd.losses import MaskedCodeLoss, UnmaskedCodeLoss
    c = TC.CODE_CASES[name]
    pred, gt, mask = TC.code_inputs(c)
    p = pred.to(DEV).requires_grad_(True)
    gt_d = gt.to(DEV)
    gt_full = torch.zeros(c["B"], c["gt_rows"], c["N"], device=DEV)
    gt_full[:, :c["nb"]] = gt_d
    gt_view = gt_full[:, :c["nb"]]                                          # batch-strided view, train.py:312-313
    if c["masked"]:
        loss = MaskedCodeLoss(c["type"])(p, gt_view, mask.to(DEV))
    else:
        loss = UnmaskedCodeLoss(c["type"])(p, gt_view)
    (loss * 1.0).backward()
    np.testing.assert_allclose(loss.item(), G["code_%s_loss" % name], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(p.grad.cpu().numpy(), G["code_%s_grad" % name], rtol=2e-5, atol=1e-8)
    ol, og = TO.code_loss(pred.numpy(), gt.numpy(), None if mask is None else mask.numpy(), c["type"])
    np.testing.assert_allclose(loss.item(), ol, rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(p.grad.cpu().numpy(), og, rtol=2e-5, atol=1e-8)


def test_code_loss_slices_of_logit_block_and_weighting():
    """train.py:307-318: the three predictions are slices of one (B,13,N) block; a weighted sum back-propagates."""
    from checkerpose_amd.losses import MaskedCodeLoss, UnmaskedCodeLoss
    B, N = 5, 512
    bits = (TC.det_tensor((B, 13, N), 77) * 3).to(DEV).requires_grad_(True)
    gtx = (TC.det_tensor((B, 16, N), 78) > 0).float().to(DEV)
    gtr = (TC.det_tensor((B, 1, N), 79) > 0).float().to(DEV)
    l = UnmaskedCodeLoss("BCE")(bits[:, 0:1], gtr) + 0.5 * MaskedCodeLoss("BCE")(bits[:, 1:7], gtx[:, :6], gtr)
    l.backward()
    ref_bits = bits.detach().cpu().requires_grad_(True)
    f = torch.nn.functional.binary_cross_entropy_with_logits
    raw = f(ref_bits[:, 1:7], gtx[:, :6].cpu(), reduction="none") * gtr.cpu()
    lr = f(ref_bits[:, 0:1], gtr.cpu()) + 0.5 * raw.sum() / (gtr.cpu().sum().clamp(min=1.0) * 6)
    lr.backward()
    np.testing.assert_allclose(l.item(), lr.item(), rtol=3e-6)
    np.testing.assert_allclose(bits.grad.cpu().numpy(), ref_bits.grad.numpy(), rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("name", list(TC.CE_CASES))
def test_masked_ce_loss(name):
    """MaskedCodeLoss("CE") (code_loss.py:36-37,47-61; cp_masked_ce_loss) vs the reference-made ce_loss.npz and the oracle"""
    from checkerpose_amd.losses import MaskedCodeLoss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ce_loss.npz"))
    pred, gt, mask = TC.ce_inputs(*TC.CE_CASES[name])
    p = pred.to(DEV).requires_grad_(True)
    loss = MaskedCodeLoss("CE")(p, gt.to(DEV), mask.to(DEV))
    (loss * 1.0).backward()
    np.testing.assert_allclose(loss.item(), g[name + "_loss"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g[name + "_grad"], rtol=2e-5, atol=1e-8)
    ol, og = TO.masked_ce_loss(pred.numpy(), gt.numpy(), mask.numpy())
    np.testing.assert_allclose(loss.item(), ol, rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(p.grad.cpu().numpy(), og, rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("name", list(TC.MASK_CASES))
def test_mask_loss(name):
    from checkerpose_amd.losses import MaskLoss_interpolate
    c = TC.MASK_CASES[name]
    pred, gt = TC.mask_inputs(c)
    p = pred.to(DEV).requires_grad_(True)
    loss = MaskLoss_interpolate()(p[:, c["ch"]:c["ch"] + 1], gt.to(DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), G["mask_%s_loss" % name], rtol=3e-6)
    np.testing.assert_allclose(p.grad.cpu().numpy(), G["mask_%s_grad" % name], rtol=2e-5, atol=1e-10)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", list(TC.EDGE_CASES))
def test_edgeconv_forward_backward(name, dtype):
    from checkerpose_amd.train_ops import edgeconv_aggregate
    c = TC.EDGE_CASES[name]
    x, idx, gup = TC.edge_inputs(c)
    wpq, sc, sh = TC.edge_folded_weights(c)
    xd = x.to(DEV).requires_grad_(True)
    pq32 = torch.einsum("bcn,oc->bno", xd, wpq.to(DEV)) * sc.to(DEV) + sh.to(DEV)
    pq = pq32.to(dtype)
    out = edgeconv_aggregate(pq, idx.to(DEV), c["slope"])
    out.backward(gup.to(DEV).transpose(1, 2).to(dtype))
    if dtype == torch.float32:
        np.testing.assert_allclose(out.detach().cpu().numpy().transpose(0, 2, 1), G["edge_%s_out" % name], rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), G["edge_%s_dx" % name], rtol=1e-4, atol=1e-5)
    # against the oracle on exactly the (rounded) pq the kernel saw -- tight in both storage types
    pq_np = pq.detach().float().cpu().numpy()
    g_np = gup.transpose(1, 2).to(dtype).float().numpy()
    dpq_o = TO.edgeconv_gather_max_bwd(pq_np, idx.numpy(), g_np, c["slope"])
    pq_leaf = pq.detach().clone().requires_grad_(True)
    edgeconv_aggregate(pq_leaf, idx.to(DEV), c["slope"]).backward(gup.to(DEV).transpose(1, 2).to(dtype))
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)   # bf16: output rounding only
    np.testing.assert_allclose(pq_leaf.grad.float().cpu().numpy(), dpq_o, **tol)


def test_edgeconv_backward_full_size_properties():
    """B=64, N=512, K=20, C=256 (the pipeline's EdgeConv): every (i,c) gradient lands on exactly one neighbour, so
    sum_j dP'[b,j,c] == sum_i dQ'[b,i,c]; the reverse-graph gather is bit-reproducible; a winner is a neighbour."""
    from checkerpose_amd.train_ops import edgeconv_aggregate, reverse_graph
    B, N, K, C = 64, 512, 20, 256
    pts = TC.det_tensor((1, 3, N), 5)
    idx = (-((pts[:, :, :, None] - pts[:, :, None, :]) ** 2).sum(1)).topk(K, dim=-1)[1][0].to(DEV)
    rev = reverse_graph(idx.int())
    pq = TC.det_tensor((B, N, 2 * C), 6).to(DEV).requires_grad_(True)
    g = TC.det_tensor((B, N, C), 7).to(DEV)
    edgeconv_aggregate(pq, idx, 0.2, rev=rev).backward(g)
    d1 = pq.grad.clone()
    pq.grad = None
    edgeconv_aggregate(pq, idx, 0.2, rev=rev).backward(g)
    assert torch.equal(d1, pq.grad)
    sP, sQ = d1[:, :, :C].double().sum(1), d1[:, :, C:].double().sum(1)
    np.testing.assert_allclose(sP.cpu().numpy(), sQ.cpu().numpy(), rtol=1e-5, atol=1e-4)
    # nodes nobody points to get exactly zero
    indeg = torch.bincount(idx.flatten(), minlength=N)
    assert torch.all(d1[:, indeg == 0, :C] == 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_index2feat_forward_backward(dtype):
    from checkerpose_amd.train_ops import index2feat_gather
    c = TC.I2F_CASES["k2"]
    patches, x_id, y_id, mask, gup = TC.i2f_inputs(c)
    pd = patches.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype).requires_grad_(True)     # channels-last
    out = index2feat_gather(pd, x_id.to(DEV), y_id.to(DEV), mask.to(DEV), c["k"])
    out.backward(gup.to(DEV).to(dtype))
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), G["i2f_k2_out"], **tol)
    np.testing.assert_allclose(pd.grad.float().cpu().numpy().transpose(0, 3, 1, 2), G["i2f_k2_dpatches"], **tol)


def test_index2feat_backward_full_size_conservation():
    """B=64, N=512, 65x65x64 patch map: sum over pixels of dpatches == sum over keypoints/taps of gout*mask."""
    from checkerpose_amd.train_ops import index2feat_gather
    B, N, E, Hp = 64, 512, 64, 65
    pd = TC.det_tensor((B, Hp, Hp, E), 8).to(DEV).requires_grad_(True)
    u = (TC.det_tensor((2, B, N), 9) * 0.5 + 0.5).clamp(0, 0.999)
    x_id, y_id = (u[0] * 32).long().to(DEV), (u[1] * 32).long().to(DEV)
    mask = (TC.det_tensor((B, N), 10) > 0).float().to(DEV)
    g = TC.det_tensor((B, N, 4 * E), 11).to(DEV)
    index2feat_gather(pd, x_id, y_id, mask, 2).backward(g)
    want = (g.double() * mask[:, :, None].double()).reshape(B, N, 4, E).sum((1, 2))
    np.testing.assert_allclose(pd.grad.double().sum((1, 2)).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_train_ops_reject_bad_arguments(lib):
    assert lib.cp_code_loss(None, 2, 1, 10, 1, 10, None, 1, 1, 10, 1, None, 10, 1) < 0          # CE not built
    assert lib.cp_code_loss(None, 0, None, 10, 1, 10, None, 1, 1, 10, 1, None, 10, 1) < 0       # null pred
    assert lib.cp_mask_loss(None, 1, 10, 1, 1, 64, 64, 64, 64, 1, None, 10, 1) < 0              # batch stride < h*w
    assert lib.cp_edgeconv_gather_max_bwd(None, 0, None, None, None, None, None, None, None, None, 1, 1, 1, 4, 1, 4, 0, 0.2) < 0
    assert lib.cp_index2feat_gather_bwd(None, None, None, None, None, None, 1, 1, 1, 1, 4, 2, 16, 0) < 0


def _opt_problem(seed=0):
    """parameter tensors of assorted sizes (incl. > one block, odd sizes) whose gradients are slices of ONE flat buffer at odd
    offsets -- exactly how the training program hands them out (4-byte aligned only)"""
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 3, 3), (18,), (5000,), (36, 18, 3, 3), (1,), (2049,), (144, 144, 3, 3)]
    params = [torch.randn(s, generator=g) for s in shapes]
    n = sum(p.numel() for p in params)
    grads = [torch.randn(5, n + 3, generator=g) * 0.1 for _ in range(1)][0]
    return shapes, params, grads


@pytest.mark.parametrize("kind", ["adam", "adam_wd", "sgd_mom", "sgd_plain_wd"])
def test_multi_tensor_optimizers_match_torch(kind):
    """checkerpose_amd.optim.Adam / SGD (ONE launch over all tensors) against torch.optim on the same parameters and gradients for 5
    steps; the last tensor receives its first gradient at step 3 (torch counts steps per parameter); state_dict round trip after 3
    steps into a fresh optimizer continues identically and loads into torch's own optimizer."""
    from checkerpose_amd import optim as O
    shapes, p0, grads = _opt_problem()
    mk = {"adam": (lambda ps: O.Adam(ps, lr=2e-3), lambda ps: torch.optim.Adam(ps, lr=2e-3)),
          "adam_wd": (lambda ps: O.Adam(ps, lr=1e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.05),
                      lambda ps: torch.optim.Adam(ps, lr=1e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.05)),
          "sgd_mom": (lambda ps: O.SGD(ps, lr=0.05, momentum=0.9), lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9)),
          "sgd_plain_wd": (lambda ps: O.SGD(ps, lr=0.05, weight_decay=0.01), lambda ps: torch.optim.SGD(ps, lr=0.05, weight_decay=0.01))}[kind]

    def run(make, steps, resume_from=None, resume_make=None):
        ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in p0]
        opt = make(ps)
        flat = torch.zeros(grads.shape[1], device=DEV)
        sd = None
        for t in range(steps):
            flat.copy_(grads[t].to(DEV))
            off = 1                                                          # odd offset: views are 4-byte aligned only
            for i, p in enumerate(ps):
                p.grad = flat[off:off + p.numel()].view_as(p) if (i + 1 < len(ps) or t >= 2) else None
                off += p.numel()
            opt.step()
            if resume_from is not None and t + 1 == resume_from:
                sd = opt.state_dict()
                opt = (resume_make or make)(ps)
                opt.load_state_dict(sd)
        torch.cuda.synchronize()
        return [p.detach().cpu() for p in ps], opt

    want, _ = run(mk[1], 5)
    got, opt = run(mk[0], 5)
    for a, b, s in zip(got, want, shapes):
        assert float((a - b).abs().max()) <= 2e-6 * (float(b.abs().max()) + 1e-6), s
    got2, _ = run(mk[0], 5, resume_from=3)
    for a, b in zip(got2, got):
        assert torch.equal(a, b)
    got3, _ = run(mk[0], 5, resume_from=3, resume_make=mk[1])               # our state loads into torch's optimizer
    for a, b, s in zip(got3, want, shapes):
        assert float((a - b).abs().max()) <= 2e-6 * (float(b.abs().max()) + 1e-6), s
    sd = opt.state_dict()
    steps = sorted({int(v["step"]) for v in sd["state"].values()})
    assert steps == [3, 5] and "_cp_table" not in sd["param_groups"][0]
    with pytest.raises(RuntimeError):
        bad = torch.nn.Parameter(torch.zeros(4, device=DEV, dtype=torch.float64))
        bad.grad = torch.zeros_like(bad)
        O.Adam([bad]).step()
