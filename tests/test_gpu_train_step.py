"""GPU (-m gpu): one full training step of the drop-in modules (train-mode forward with batch-statistics BatchNorm +
backward through the HIP training program) against torch-CPU autograd over the oracle restatement
in train mode (oracle/checkerpose_oracle.py under bn_train()).
Tolerance (fp32 path): logits 2e-4 absolute; parameter gradients as documented in _compare() (5e-4 of each tensor's max,
1e-4 relative L2 overall, with the discontinuous choices pinned to the device's); running statistics 1e-4.  ids must be equal (otherwise the gather positions differ and the comparison is void).
The max over the K neighbours routes its gradient to one of them and a (Leaky)ReLU switches its derivative at 0 -- both
discontinuous: the oracle is teacher-forced to the device's arg-max slots (O.FORCE_KSTAR) and activation branches
(O.FORCE_MASK, read from the program's live activations between forward and backward), as the eval-mode tests teacher-force
the bit decisions; without that, the ~1e-6 forward differences flip ~1e-4 of the near-ties / near-zeros and move individual
gradient tensors by percents (measured 4e-2 rel-L2 on a 128-pixel BatchNorm bias).
"""
import pytest
import torch

from oracle import checkerpose_oracle as O
from tests.common import build_net, det_image, det_tensor, oracle_kwargs

pytestmark = pytest.mark.gpu


def _device_kstar(net, prefix=""):
    """arg-max neighbour slots the device chose in its last train-mode forward, per EdgeConv layer: {oracle prefix: (B,C',N)}"""
    pr = list(net._train_programs.values())[-1]
    out = {}
    for pfx, d in pr["prog"].debug.items():
        ks = d["st"]["kstar"]
        B, N = d["out"].B, d["out"].W
        out[prefix + pfx] = ks.view(B, N, -1).permute(0, 2, 1).long().cpu().contiguous()
    return out


def _device_masks(net, prefix=""):
    """branch every (Leaky)ReLU element took on the device ({oracle key: bool NCHW / (B,C,N) / (B,N,C) tensor}); read between
    the forward and the backward, while the activations are still live in the program's workspace"""
    pr = list(net._train_programs.values())[-1]
    prog = pr["prog"]
    out = {}
    for key, a in prog.kinks.items():
        if a.tbuf not in prog.grads:                  # off the gradient path (e.g. InitNet alone: the incre heads of branches 0-2 and what only
            continue                                  # they read): nothing keeps the buffer behind the forward, it has been recycled by now
        v = prog.read_act(a) > 0                      # (B,H,W,C)
        if a.H == 1 and "pre_query_block." in key:    # EdgeConv output: oracle layout (B,C',N)
            out[prefix + key] = v[:, 0].permute(0, 2, 1).contiguous()
        elif a.H == 1:                                # per-keypoint MLP: oracle layout (B,N,C)
            out[prefix + key] = v[:, 0].contiguous()
        else:
            out[prefix + key] = v.permute(0, 3, 1, 2).contiguous()
    return out


def _oracle_step(net, img, seeds, stage=None, init_only=False, kstar=None, masks=None, backbone="hrnet_w18", okw=None):
    O.FORCE_KSTAR.clear()
    O.FORCE_KSTAR.update(kstar or {})
    O.FORCE_MASK.clear()
    O.FORCE_MASK.update(masks or {})
    try:
        return _oracle_step_(net, img, seeds, stage, init_only, backbone, okw)
    finally:
        O.FORCE_KSTAR.clear()
        O.FORCE_MASK.clear()


def _oracle_step_(net, img, seeds, stage=None, init_only=False, backbone="hrnet_w18", okw=None):
    sd = {}
    for k, v in net.state_dict().items():
        sd[k] = v.detach().clone()
    params = [k for k, _ in net.named_parameters()]
    for k in params:
        sd[k].requires_grad_(True)
    with torch.enable_grad(), O.bn_train():
        if init_only:
            out, _, _ = O.init_net_forward(sd, "", img, net.knn_idx, net.npoint, backbone, 2, 0.2)
            outs = [out]
            ids = None
        else:
            (roi, xb, yb, seg, x_id, y_id), _ = O.posenet_forward(sd, img, net.init_net.knn_idx, net.npoint, stage=stage,
                                                                  **dict(oracle_kwargs(), **(okw or {})))
            outs = [roi, xb, yb, seg]
            ids = (x_id, y_id)
        grads = torch.autograd.grad(outs, [sd[k] for k in params], seeds, allow_unused=True)
    return outs, ids, dict(zip(params, grads)), sd


def _compare(net, ref_grads, sd_ref, tol_max=5e-4, tol_global=1e-4):
    """Gradient parity with every discontinuous choice pinned to the device's (bit decisions asserted equal, EdgeConv
    arg-max slots and (Leaky)ReLU branches teacher-forced into the oracle): every gradient tensor within 5e-4 of the oracle's
    in the max norm (relative to the tensor's largest entry), all gradients together within 1e-4 in relative L2.
    Measured on MI355X: 3.7e-5 / 6.7e-6.  (Unforced, ~1e-4 of the near-ties / near-zero pre-activations take the other
    branch under the ~1e-6 forward differences and move individual small tensors by percents: 4e-2 rel-L2 measured.)"""
    worst, num, den = (0.0, None), 0.0, 0.0
    for k, p in net.named_parameters():
        g_ref = ref_grads[k]
        g = p.grad
        if g_ref is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, "no gradient for %s" % k
        d = (g.cpu().double() - g_ref.double())
        num, den = num + float((d * d).sum()), den + float((g_ref.double() ** 2).sum())
        em = float(d.abs().max()) / max(float(g_ref.abs().max()), 1e-12)
        if em > worst[0]:
            worst = (em, k)
    glob = (num / den) ** 0.5
    print("gradient parity: global rel-L2 %.3e, worst tensor max-norm err %.3e (%s)" % (glob, worst[0], worst[1]))
    assert worst[0] <= tol_max, "gradient of %s: max err %.3e > %.1e" % (worst[1], worst[0], tol_max)
    assert glob <= tol_global, "all gradients: rel L2 err %.3e > %.1e" % (glob, tol_global)
    bufs = dict(net.named_buffers())
    for k in bufs:                                    # every BatchNorm's running statistics and step counter
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert float((bufs[k].cpu() - sd_ref[k]).abs().max()) <= 1e-4 * (1 + float(sd_ref[k].abs().max())), k
        if k.endswith("num_batches_tracked"):
            assert int(bufs[k]) == int(sd_ref[k]), k
    return worst


@pytest.mark.parametrize("stage", [None, 1])
def test_posenet_train_step_vs_oracle_autograd(stage):
    torch.manual_seed(0)
    B = 2
    net = build_net(seed=3).train()
    img = det_image(B, seed=11)   # decision margin 8.8e-4 in train mode (searched over seeds 0..23)
    active = 3 if stage is None else stage
    seeds = [det_tensor("g_roi", (B, 1, 512)), det_tensor("g_x", (B, 3 + active, 512)), det_tensor("g_y", (B, 3 + active, 512)),
             det_tensor("g_seg", (B, 2, 8 << active, 8 << active), 0.05)]
    net_cpu = build_net(seed=3).train()
    net = net.cuda()
    with torch.enable_grad():
        res = net(img.cuda(), None, stage)
        torch.cuda.synchronize()
        masks = _device_masks(net)
        torch.autograd.backward(list(res[:4]), [s.cuda() for s in seeds])
    torch.cuda.synchronize()
    outs, ids, ref_grads, sd_ref = _oracle_step(net_cpu, img, seeds, stage=stage, kstar=_device_kstar(net), masks=masks)
    for a, b in zip(res[:4], outs):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-4
    assert torch.equal(res[4].cpu(), ids[0]) and torch.equal(res[5].cpu(), ids[1]), "discrete ids differ: comparison void"
    _compare(net, ref_grads, sd_ref)
    # eval after a train step must see the updated running statistics (stale folded-BN caches dropped)
    net.eval()
    with torch.no_grad():
        ev = net(img.cuda(), None, stage)
    sd_eval = {k: v.detach() for k, v in sd_ref.items()}
    with torch.no_grad():
        ref_ev, _ = O.posenet_forward(sd_eval, img, net.init_net.knn_idx, net.npoint, stage=stage, **oracle_kwargs())
    for a, b in zip(ev[:4], ref_ev[:4]):
        assert float((a.cpu() - b).abs().max()) <= 2e-4


def test_posenet_train_step_without_graph_modules_vs_oracle_autograd():
    """`num_graph_module = 0` in InitNet and in every refinement stage (config/lm/*_woEdgeConv.txt) through the TRAINING program: the
    conv1x1 rows / the pre-graph MLP rows are written straight into the next stage's input rows, and their gradients come back
    through those slices.  Seeds searched on the CPU oracle for a train-mode decision margin (1.35e-4)."""
    B = 2
    net = build_net(seed=10, init_graph=0, graph=0).train()
    net_cpu = build_net(seed=10, init_graph=0, graph=0).train()
    img = det_image(B, seed=5)
    seeds = [det_tensor("g_roi", (B, 1, 512)), det_tensor("g_x", (B, 6, 512)), det_tensor("g_y", (B, 6, 512)), det_tensor("g_seg", (B, 2, 64, 64), 0.05)]
    net = net.cuda()
    with torch.enable_grad():
        res = net(img.cuda(), None, None)
        torch.cuda.synchronize()
        masks = _device_masks(net)
        torch.autograd.backward(list(res[:4]), [s.cuda() for s in seeds])
    torch.cuda.synchronize()
    outs, ids, ref_grads, sd_ref = _oracle_step(net_cpu, img, seeds, kstar=_device_kstar(net), masks=masks, okw=dict(init_n_graph=0, n_graph=0))
    for a, b in zip(res[:4], outs):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-4
    assert torch.equal(res[4].cpu(), ids[0]) and torch.equal(res[5].cpu(), ids[1]), "discrete ids differ: comparison void"
    _compare(net, ref_grads, sd_ref)


def test_initnet_train_step_vs_oracle_autograd():
    B = 2
    net = build_net(seed=2, full=False).train()
    img = det_image(B, seed=6)
    seeds = [det_tensor("g_init", (B, 7, 512))]
    net_cpu = build_net(seed=2, full=False).train()
    net = net.cuda()
    with torch.enable_grad():
        out = net(img.cuda())
        torch.cuda.synchronize()
        masks = _device_masks(net)
        out.backward(seeds[0].cuda())
    torch.cuda.synchronize()
    outs, _, ref_grads, sd_ref = _oracle_step(net_cpu, img, seeds, init_only=True, kstar=_device_kstar(net), masks=masks)
    assert float((out.detach().cpu() - outs[0].detach()).abs().max()) <= 2e-4
    _compare(net, ref_grads, sd_ref)


_LOOP_LOSSES = {}          # (dtype, optimizer) -> loss trajectory, for the torch-vs-HIP optimizer comparison


@pytest.mark.parametrize("dt,optim", [("fp32", "torch"), ("bf16", "torch"), ("fp32", "hip"), ("bf16", "hip_sgd")])
def test_training_loop_like_train_py_reduces_the_loss(dt, optim):
    """The step sequence of reference train.py:300-320 (zero_grad, net(data, p3d, stage), roi / x / y code losses + the two
    seg mask losses, backward, Adam step) on one fixed synthetic batch: the loss must go down.  Runs the fp32 and the
    bf16 storage program (the bf16 forward differs from fp32 in ~5 % of the bits on these random-init weights, so its
    gradients are judged by what they are for -- descent -- not by a distance to the fp32 ones).  optim: torch's Adam, or the
    one-launch drop-ins of checkerpose_amd.optim (train.py:244-246: Adam, or SGD with momentum 0.9) -- the fp32 / "hip" run must
    follow the fp32 / "torch" trajectory (same arithmetic; the gradients carry atomics noise only)."""
    from checkerpose_amd.losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
    from checkerpose_amd.losses.mask_loss import MaskLoss_interpolate
    B, N = 4, 512
    torch.manual_seed(0)
    net = build_net(seed=3).cuda().train()
    net.set_compute_dtype(dt)
    img = det_image(B, seed=7).cuda()
    roi_gt = (det_tensor("t_roi", (B, 1, N)) > -0.5).float().cuda()
    x_gt = (det_tensor("t_x", (B, 16, N)) > 0).float().cuda()
    y_gt = (det_tensor("t_y", (B, 16, N)) > 0).float().cuda()
    m_vis = (det_tensor("t_mv", (B, 128, 128)) > 0).float().cuda()
    m_full = (det_tensor("t_mf", (B, 128, 128)) > -0.3).float().cuda()
    roi_loss, bit_loss, seg_loss = UnmaskedCodeLoss("BCE"), MaskedCodeLoss("BCE"), MaskLoss_interpolate()
    from checkerpose_amd import optim as hip_optim
    opt = {"torch": lambda: torch.optim.Adam(net.parameters(), lr=2e-4), "hip": lambda: hip_optim.Adam(net.parameters(), lr=2e-4),
           "hip_sgd": lambda: hip_optim.SGD(net.parameters(), lr=2e-3, momentum=0.9)}[optim]()
    p3d = net.init_net.knn_idx.new_zeros(1, 3, N).float().cuda().expand(B, -1, -1)
    losses = []
    with torch.enable_grad():
        for it in range(8):
            opt.zero_grad()
            roi, xb, yb, seg, _, _ = net(img, p3d, 3)
            nb = xb.shape[1]
            loss = roi_loss(roi, roi_gt) + bit_loss(xb, x_gt[:, :nb], roi_gt) + bit_loss(yb, y_gt[:, :nb], roi_gt) \
                + seg_loss(seg[:, 0:1], m_vis) + seg_loss(seg[:, 1:2], m_full)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    print(dt, optim, "losses", ["%.4f" % v for v in losses])
    assert all(v == v for v in losses), "NaN loss"
    assert losses[-1] < 0.9 * losses[0], losses
    assert min(losses[4:]) < min(losses[:2]), losses
    _LOOP_LOSSES[(dt, optim)] = losses
    if (dt, optim) == ("fp32", "hip") and ("fp32", "torch") in _LOOP_LOSSES:
        # Adam's first steps are sign-like (m / sqrt(v) ~ +-1): the run-to-run atomics noise of the gradients is amplified step by step
        # (measured 1e-7, 1e-7, 2e-4 .. 1.3e-3, 1e-4, 3e-3 relative over several runs of the SAME build: the third loss already sits behind
        # two sign-like updates); the two optimizers themselves agree to 2e-6 (test_gpu_train_ops.py)
        for k, (a, b) in enumerate(zip(losses, _LOOP_LOSSES[("fp32", "torch")])):
            assert abs(a - b) <= (1e-3 if k < 2 else 2e-2) * abs(b), (losses, _LOOP_LOSSES[("fp32", "torch")])



def test_resnet34_initnet_train_step_vs_oracle_autograd():
    """ResNet-34 backbone (config/lm/res34GNN2_*): 7x7/s2 stem, max-pool backward, stride-2 BasicBlocks with 1x1/s2 shortcuts"""
    B = 2
    net = build_net(seed=5, full=False, backbone="resnet34").train()
    net_cpu = build_net(seed=5, full=False, backbone="resnet34").train()
    img = det_image(B, seed=3)
    seeds = [det_tensor("g_init34", (B, 7, 512))]
    net = net.cuda()
    with torch.enable_grad():
        out = net(img.cuda())
        torch.cuda.synchronize()
        masks = _device_masks(net)
        out.backward(seeds[0].cuda())
    torch.cuda.synchronize()
    outs, _, ref_grads, sd_ref = _oracle_step(net_cpu, img, seeds, init_only=True, kstar=_device_kstar(net), masks=masks,
                                              backbone="resnet34")
    assert float((out.detach().cpu() - outs[0].detach()).abs().max()) <= 2e-4
    _compare(net, ref_grads, sd_ref)


@pytest.mark.parametrize("backbone", ["hrnet_w18_small", "hrnet_w30"])
def test_other_hrnet_initnet_train_step_vs_oracle_autograd(backbone):
    """hrnet_w18_small / hrnet_w30 (tests/test_gpu_parity.py: test_e2e_other_hrnet_backbones) through the TRAINING program: one
    32-plane Bottleneck in layer1 / two BasicBlocks per branch / one module per stage, other widths -- every gradient vs autograd over
    the train-mode oracle"""
    B = 2
    net = build_net(seed=5, full=False, backbone=backbone).train()
    net_cpu = build_net(seed=5, full=False, backbone=backbone).train()
    img = det_image(B, seed=3)
    seeds = [det_tensor("g_init_" + backbone, (B, 7, 512))]
    net = net.cuda()
    with torch.enable_grad():
        out = net(img.cuda())
        torch.cuda.synchronize()
        masks = _device_masks(net)
        out.backward(seeds[0].cuda())
    torch.cuda.synchronize()
    outs, _, ref_grads, sd_ref = _oracle_step(net_cpu, img, seeds, init_only=True, kstar=_device_kstar(net), masks=masks, backbone=backbone)
    assert float((out.detach().cpu() - outs[0].detach()).abs().max()) <= 2e-4
    _compare(net, ref_grads, sd_ref)


def test_lm_twin_training_loop_per_sample_graphs():
    """LM networks (pipeline_lm.py): per-sample kNN graphs (obj_ids) through the train-mode EdgeConv kernels; the loss falls"""
    from checkerpose_amd.losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
    B, N = 4, 512
    net = build_net(seed=3, lm=True).cuda().train()
    img = det_image(B, seed=9).cuda()
    obj_ids = torch.tensor([1, 5, 9, 15])
    roi_gt = (det_tensor("l_roi", (B, 1, N)) > -0.5).float().cuda()
    x_gt = (det_tensor("l_x", (B, 16, N)) > 0).float().cuda()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    p3d = torch.zeros(B, 3, N).cuda()
    losses = []
    with torch.enable_grad():
        for it in range(5):
            opt.zero_grad()
            roi, xb, yb, seg, _, _ = net(img, p3d, obj_ids, 2)
            loss = UnmaskedCodeLoss("BCE")(roi, roi_gt) + MaskedCodeLoss("BCE")(xb, x_gt[:, :xb.shape[1]], roi_gt) \
                + MaskedCodeLoss("BCE")(yb, x_gt[:, :yb.shape[1]], roi_gt)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    print("lm losses", ["%.4f" % v for v in losses])
    assert all(v == v for v in losses) and losses[-1] < 0.9 * losses[0], losses


def test_train_program_follows_reassigned_parameter_storage():
    """the training program reads the LIVE parameter storage; re-assigning a parameter's tensor must not leave it on stale
    pointers: same forward as a fresh net with the new values"""
    B = 2
    img = det_image(B, seed=7).cuda()
    net = build_net(seed=3).cuda().train()
    with torch.no_grad():
        a = net(img, None, 1)
        w = net.seg_block.bias                                 # (a conv weight in front of a train-mode BatchNorm would be
        w.data = (w.data + 1.0).clone()                        #  normalised away) -- new storage
        b = net(img, None, 1)
    ref = build_net(seed=3).cuda().train()
    with torch.no_grad():
        ref.seg_block.bias.add_(1.0)
        c = ref(img, None, 1)
    assert float((a[3] - b[3]).abs().max()) > 1e-3          # the change is visible
    assert float((b[3] - c[3]).abs().max()) <= 1e-5 and float((b[0] - c[0]).abs().max()) <= 1e-5


def test_train_mode_accepts_uint8_crops():
    """row N3 in train mode: raw uint8 HWC crops are normalised on the device; same logits / gradients as the fp32 crops"""
    B = 2
    u8 = (det_image(B, seed=4) * 40 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    f32 = O.preprocess_uint8(u8)
    seeds = [det_tensor("g_roi", (B, 1, 512)).cuda(), det_tensor("g_x", (B, 4, 512)).cuda(), det_tensor("g_y", (B, 4, 512)).cuda(),
             det_tensor("g_seg", (B, 2, 16, 16), 0.05).cuda()]
    grads = []
    for inp in (u8, f32):
        net = build_net(seed=3).cuda().train()
        with torch.enable_grad():
            res = net(inp.cuda(), None, 1)
            torch.autograd.backward(list(res[:4]), seeds)
        grads.append((res[1].detach(), net.up_net[0][3].weight.grad.clone(), net.init_net.img_backbone.conv1.weight.grad.clone()))
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-5 * (1 + float(b.abs().max()))


@pytest.mark.parametrize("seed", [1, 4])
def test_bf16_contract_on_trained_like_weights(seed):
    """checkerpose_amd/trained_like.py: 300 steps of the HIP training program (train.py:300-320's step sequence, bf16, B = 32) on the
    synthetic translation task -- the loss must fall, the network must generalise to held-out crops -- then the bf16 eval path (the
    bench's kernel selection: keypoint side in IEEE half) against the fp32 eval path of the SAME trained weights.  Training runs in
    the deterministic mode (cp_set_deterministic), so the trained network -- and every statistic below -- is the same in every run.
    Asserted with the contract's OWN clauses (agreement.margin_contract_violations == [], no slack) and its floors: teacher-forced
    rows >= 98 %, seg >= 99 %, mean |dlogit| <= 2 % of the logit RMS; free-running rows >= 95 %, final id pairs >= 90 %.  Recorded:
    profiles/r06_trained_like_*.json (seeds 1-5 at 300 steps and seed 1 at 3 000 steps: no violation in any; seed 4 is the run
    with the lowest free-running agreement).  Seeds 6-15 were run afterwards: three of them miss a tail clause by one flip -- see
    test_bf16_contract_known_marginal_seed below and DESIGN.md section 7."""
    import checkerpose_amd
    from checkerpose_amd.trained_like import train_then_measure
    checkerpose_amd.set_deterministic(True)
    try:
        r = train_then_measure(npoint=512, steps=300, batch=32, lr=5e-4, seed=seed, held_out=8)
    finally:
        checkerpose_amd.set_deterministic(False)
    ls = r["loss_every_25_steps"]
    assert ls[-1] < 0.5 * ls[0], ls
    assert r["held_out"]["roi_bit_accuracy_vs_gt"] >= 0.9, r["held_out"]
    tf, fr = r["teacher_forced"], r["free_running"]
    print("trained-like seed %d:" % seed, r["held_out"], "violations:", r["margin_contract_violations"],
          "tf max flip margin %.3f = %.2f x mean (rms %.2f)" % (tf["max_flip_margin"], tf["max_flip_margin"] / tf["mean_abs_dlogit"], tf["logit_rms"]),
          "fr id equal %.4f" % fr["xy_id_equal"])
    assert r["margin_contract_violations"] == [], r["margin_contract_violations"]
    assert tf["bit_agreement_min_row"] >= 0.98 and tf["seg_agreement"] >= 0.99 and tf["mean_abs_dlogit_over_rms"] <= 0.02, tf
    assert tf["max_abs_dlogit"] <= 0.5 and tf["flip_rate_by_margin"]["0.2-1"]["flips"] == 0 and tf["flip_rate_by_margin"]["1-inf"]["flips"] == 0, tf
    assert fr["bit_agreement_min_row"] >= 0.95 and fr["xy_id_equal"] >= 0.90 and fr["id_abs_err_mean_px"] <= 0.5, fr


def test_bf16_contract_known_marginal_seed():
    """The tail clauses (b) / (c) of the contract are statistical at this error level: of 15 training seeds
    (profiles/r06_trained_like_300steps_seeds1-15.json) three miss one of them by ONE flip.  This pins the worst of the three, seed 10
    -- deterministic training, so the same network on every box: exactly one violated clause, (b), a flip at 6.83 x the mean |dlogit|
    (margin 0.084, clause (a)'s bound is 0.2); every floor of the contract holds.  A build whose error grows shows up here as a
    second violation or a larger excess; one whose error shrinks makes this test fail on the `== 1`, which is the cue to retire it."""
    import checkerpose_amd
    from checkerpose_amd.trained_like import train_then_measure
    checkerpose_amd.set_deterministic(True)
    try:
        r = train_then_measure(npoint=512, steps=300, batch=32, lr=5e-4, seed=10, held_out=8)
    finally:
        checkerpose_amd.set_deterministic(False)
    tf, fr = r["teacher_forced"], r["free_running"]
    v = r["margin_contract_violations"]
    print("trained-like seed 10:", v, "per-row ratio %.2f" % tf["max_flip_margin_over_row_mean"])
    assert len(v) == 1 and v[0].startswith("teacher-forced: a flip at margin") and "6 x mean" in v[0], v
    assert tf["max_flip_margin"] <= 7.5 * tf["mean_abs_dlogit"] and tf["max_flip_margin"] < 0.2 and tf["flips_above_margin"] <= 2, tf
    assert tf["bit_agreement_min_row"] >= 0.98 and tf["seg_agreement"] >= 0.99 and tf["mean_abs_dlogit_over_rms"] <= 0.02 and tf["max_abs_dlogit"] <= 0.5, tf
    assert fr["bit_agreement_min_row"] >= 0.95 and fr["xy_id_equal"] >= 0.90 and fr["id_abs_err_mean_px"] <= 0.5, fr
    assert fr["id_mismatches_explained_frac"] >= 0.95 and fr["id_mismatches_self_subtau_frac"] >= 0.60, fr


def _train_steps(steps, deterministic, seed=1, batch=8):
    """`steps` steps of the trained-like task (checkerpose_amd/trained_like.py) -> every parameter and buffer, on the CPU"""
    import checkerpose_amd
    from checkerpose_amd.trained_like import train_net
    checkerpose_amd.set_deterministic(deterministic)
    try:
        net, _, _, losses = train_net(npoint=512, steps=steps, batch=batch, lr=5e-4, seed=seed)
        torch.cuda.synchronize()
        return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, losses
    finally:
        checkerpose_amd.set_deterministic(False)


def test_deterministic_training_mode_is_bit_reproducible():
    """`checkerpose_amd.set_deterministic(True)` (CHECKERPOSE_AMD_DETERMINISTIC=1, `net.deterministic = True`): two runs of the same 50
    training steps (train.py:303-320's sequence through the HIP training program: train-mode forward, five losses, backward, Adam)
    end in BIT-IDENTICAL parameters and BatchNorm buffers -- what the reference's CPU step gives for free.  BatchNorm sums go one
    block per accumulator set, every weight gradient through the fixed-order partial-tile reduction, Index2Feat's scatter as an
    ordered gather (include/checkerpose_hip.h: cp_set_deterministic)."""
    a, la = _train_steps(50, True)
    b, lb = _train_steps(50, True)
    assert la == lb, (la, lb)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, "%d of %d tensors differ between two deterministic runs, e.g. %s" % (len(bad), len(a), bad[:5])
    assert la[-1] < la[0]


def test_deterministic_mode_matches_default_mode_numerically():
    """the deterministic mode changes the ORDER of the accumulations only: after 3 steps its parameters agree with the default
    mode's to rounding (the default mode's own run-to-run spread is of the same size)"""
    a, _ = _train_steps(3, True)
    b, _ = _train_steps(3, False)
    worst = max(float((a[k].float() - b[k].float()).abs().max()) / (1e-6 + float(b[k].float().abs().max())) for k in a
                if a[k].dtype.is_floating_point)
    assert worst <= 2e-2, worst
