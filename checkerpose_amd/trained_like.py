"""bf16 accuracy contract on TRAINED-LIKE weights.

Every other bf16 statistic of this repository is measured on closed-form random-init weights (no checkpoint of the reference is
available offline, no dataset either: SURVEY.md section 8c).  A trained network differs where the contract looks: its BatchNorm
running statistics are real batch statistics, its logits have margins (the losses push them away from zero), its weights are
correlated.  This module makes such a network with the repository's OWN training step -- the step sequence of the reference's
`train.py:300-320` (zero_grad, train-mode forward, the five losses of `train.py:310-318`, backward, Adam) through the HIP training
program -- on a synthetic task whose targets ARE a function of the image, so that it generalises to held-out crops:

  a crop is a smooth periodic pattern translated by (u, v) in [0, 1)^2; keypoint n's ground-truth pixel on the 64 x 64 code grid is
  its canonical position (from the object's normalised FPS keypoint, x / y coordinates) translated by the same (u, v); the 6 + 6
  code bits are that pixel's binary digits MSB first (`binary_code_helper/class_id_encoder_decoder.py:88` convention, mirrored by
  `pipeline.py:72-82`), the RoI bit says whether it falls inside the crop's inner window, the two masks are a disc (full) and its
  left part (visible) that move with (u, v).

`train_then_measure` trains for `steps` steps (bf16 training program, B crops per step, fresh (u, v) every step), then measures
`agreement.logit_agreement` of the bf16 eval path against the fp32 eval path of the SAME trained weights on held-out crops,
teacher-forced and free-running, plus the margin clauses.  Needs a GPU (the training program has no CPU fallback)."""
import math

import torch


def canonical_positions(p3d_normed):
    """(1, 3, N) normalised keypoints -> (N, 2) canonical code-grid positions in [12, 52) (x from coordinate 0, y from coordinate 1)"""
    xy = p3d_normed[0, :2].t().float()                       # (N, 2) in [-1, 1]
    return 32.0 + 20.0 * xy.clamp(-1.0, 1.0)


def make_batch(B, pos, gen, device):
    """one synthetic batch: img (B,3,256,256) f32 ~ unit variance, roi_gt (B,1,N), x_gt / y_gt (B,6,N) bit planes MSB first,
    m_vis / m_full (B,64,64); everything a closed-form function of the per-crop translation (u, v) drawn from `gen`"""
    uv = torch.rand(B, 2, generator=gen, device=device)                              # (B, 2)
    ys, xs = torch.meshgrid(torch.arange(256, device=device, dtype=torch.float32), torch.arange(256, device=device, dtype=torch.float32),
                            indexing="ij")
    img = torch.empty(B, 3, 256, 256, device=device)
    for c, (fx, fy, ph) in enumerate(((1.0, 1.0, 0.0), (2.0, 1.0, 0.7), (1.0, 2.0, 1.9))):
        ax = 2.0 * math.pi * fx * (xs[None] / 256.0 - uv[:, 0, None, None])
        ay = 2.0 * math.pi * fy * (ys[None] / 256.0 - uv[:, 1, None, None])
        img[:, c] = 2.0 * torch.sin(ax + ph) * torch.cos(ay)                         # variance 1
    img += 0.1 * torch.randn(B, 3, 256, 256, generator=gen, device=device)
    shift = 24.0 * (uv - 0.5)                                                        # +-12 code pixels
    p = pos.to(device)[None] + shift[:, None, :]                                     # (B, N, 2)
    inside = ((p >= 12.0) & (p < 52.0)).all(-1)                                      # the crop's inner window
    ids = p.clamp(0.0, 63.0).floor().long()                                          # (B, N, 2)
    bits = ((ids[..., None] >> torch.arange(5, -1, -1, device=device)) & 1).float()  # (B, N, 2, 6) MSB first
    x_gt, y_gt = bits[:, :, 0].permute(0, 2, 1).contiguous(), bits[:, :, 1].permute(0, 2, 1).contiguous()
    roi_gt = inside.float()[:, None, :].contiguous()
    gy, gx = torch.meshgrid(torch.arange(64, device=device, dtype=torch.float32), torch.arange(64, device=device, dtype=torch.float32),
                            indexing="ij")
    cx, cy = 32.0 + shift[:, 0, None, None], 32.0 + shift[:, 1, None, None]
    m_full = (((gx[None] - cx) ** 2 + (gy[None] - cy) ** 2) < 18.0 ** 2).float()
    m_vis = m_full * (gx[None] < cx + 6.0).float()
    return img, roi_gt, x_gt, y_gt, m_vis, m_full


def train_net(npoint=512, steps=300, batch=32, lr=5e-4, seed=1, device=None, log=None):
    """`steps` steps of the repository's training program (bf16) on the synthetic translation task.  Returns (net in train mode's
    final state, canonical positions, the generator the held-out crops continue from, losses every 25 steps)."""
    from .losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
    from .losses.mask_loss import MaskLoss_interpolate
    from .optim import Adam
    from .synthetic import ape_p3d, build_net
    dev = torch.device(device if device is not None else "cuda")
    N = npoint
    p3d = ape_p3d(N)
    pos = canonical_positions(p3d)
    gen = torch.Generator(device=dev).manual_seed(1234 + seed)
    was_grad = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    try:
        net = build_net(npoint=N, seed=seed).to(dev).train()
        net.set_compute_dtype("bf16")
        roi_loss, bit_loss, seg_loss = UnmaskedCodeLoss("BCE"), MaskedCodeLoss("BCE"), MaskLoss_interpolate()
        opt = Adam(net.parameters(), lr=lr)
        p3 = torch.zeros(1, 3, N, device=dev).expand(batch, -1, -1)
        losses = []
        for it in range(steps):
            img, roi_gt, x_gt, y_gt, m_vis, m_full = make_batch(batch, pos, gen, dev)
            opt.zero_grad(set_to_none=True)
            roi, xb, yb, seg, _, _ = net(img, p3, 3)
            loss = roi_loss(roi, roi_gt) + bit_loss(xb, x_gt, roi_gt) + bit_loss(yb, y_gt, roi_gt) \
                + seg_loss(seg[:, 0:1], m_vis) + seg_loss(seg[:, 1:2], m_full)
            loss.backward()
            opt.step()
            if it % 25 == 0 or it == steps - 1:
                losses.append(round(float(loss.detach()), 4))
                if log:
                    log("trained_like step %d loss %.4f" % (it, losses[-1]))
    finally:
        torch.set_grad_enabled(was_grad)
    return net, pos, gen, losses


def train_then_measure(npoint=512, steps=300, batch=32, lr=5e-4, seed=1, held_out=8, device=None, log=None, attribution=False,
                       selection="per_crop"):
    from .agreement import attribute_groups, logit_agreement, margin_contract_violations
    dev = torch.device(device if device is not None else "cuda")
    N = npoint
    net, pos, gen, losses = train_net(npoint, steps, batch, lr, seed, dev, log)
    with torch.no_grad():
        net.eval()
        if selection is not None:      # "per_crop": the launches the bench's 256-crop step is made of (the held-out batch is small, and
            net.set_kernel_selection(selection)    # "auto" would pick the small-batch kernels, whose keypoint side stays bf16)
        img, roi_gt, x_gt, y_gt, m_vis, m_full = make_batch(held_out, pos, gen, dev)
        net.set_compute_dtype("fp32")
        net.clone_outputs = True
        ref = [t.clone() for t in net(img, None)]
        z = torch.cat(ref[:3], 1)
        gt = torch.cat([roi_gt, x_gt, y_gt], 1)
        m = roi_gt.expand(-1, 12, -1)
        acc_roi = float(((ref[0] > 0).float() == roi_gt).float().mean())
        acc_bits = float((((z[:, 1:] > 0).float() == gt[:, 1:]).float() * m).sum() / m.sum().clamp_min(1.0))
        t = torch.zeros(held_out, 13, N, device=dev)
        t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
        net.set_compute_dtype("bf16")
        forced = logit_agreement(net.forward_teacher_forced(img, t), ref)
        free = logit_agreement(net(img, None), ref, tau=forced["tau"], explain=True, knn_idx=net.init_net.knn_idx)
    keep = ("bit_agreement_min_row", "bit_agreement_all_rows", "xy_id_equal", "id_abs_err_mean_px", "seg_agreement", "max_abs_dlogit",
            "mean_abs_dlogit", "logit_rms", "mean_abs_dlogit_over_rms", "tau", "tau_cap", "flips", "flips_above_margin", "max_flip_margin",
            "flip_rate_by_margin", "id_mismatches", "id_mismatches_explained_frac", "id_mismatches_self_subtau_frac",
            "perturbed_coverage_by_stage", "max_flip_margin_over_row_mean", "flips_above_4x_row_mean", "max_flip_margin_over_stage_mean",
            "flips_above_4x_stage_mean", "row_mean_abs_dlogit")
    absz = z.abs()
    return {"weights": "random init (seed %d) + %d steps of the HIP training program (bf16, B = %d, Adam lr %g) on the synthetic translation "
                       "task of checkerpose_amd/trained_like.py" % (seed, steps, batch, lr),
            "loss_every_25_steps": losses,
            "held_out": {"crops": held_out, "roi_bit_accuracy_vs_gt": round(acc_roi, 4), "code_bit_accuracy_vs_gt_inside_roi": round(acc_bits, 4),
                         "logit_abs_median": round(float(absz.median()), 4),
                         "logits_below_0.05": round(float((absz < 0.05).float().mean()), 5)},
            "vs": "fp32 HIP eval path of the same trained weights, %d held-out crops, kernel selection %r" % (held_out, selection),
            "margin_contract_violations": margin_contract_violations(forced, free),
            "teacher_forced": {k: forced[k] for k in keep if k in forced},
            "free_running": {k: free[k] for k in keep if k in free},
            **({"attribution": attribute_groups(net, img, log=log)} if attribution else {})}
