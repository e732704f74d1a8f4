"""Accuracy contract of the reduced-precision (bf16) path, as numbers: how far its outputs are from a reference run of the
same network (the fp32 HIP path, which tests/test_gpu_parity.py pins to the CPU oracle at 1e-4; or the oracle itself).

The 1e-4 criterion of north_star cannot apply to bf16 storage (8 significant bits), and ADD(-S) is not measurable offline
(no datasets / checkpoints / PnP solver), so the bf16 path is judged on what the downstream PnP stage consumes
(test.py:294-329): the thresholded bits of all 13 logit rows, the final pixel ids, the two segmentation masks, plus a
bound on the logit error itself.  Used by tests (thresholds) and by bench.py (reported in the JSON line).
"""
import torch

ROWS = ("roi", "x5", "x4", "x3", "x2", "x1", "x0", "y5", "y4", "y3", "y2", "y1", "y0")   # MSB first (pipeline.py:72-82)
_Z0 = torch.tensor([0x33C00000], dtype=torch.int32).view(torch.float32)                  # include/checkerpose_hip.h


def _bit(z):
    return z > _Z0.to(z.device)


def logit_agreement(out, ref, tau=None, explain=False, knn_idx=None, graph_ids=None, init_bits=3):
    """out / ref: 6-tuples (roi (B,1,N), x_bits (B,nx,N), y_bits (B,ny,N), seg (B,2,h,w), x_id, y_id) of the same forward.
    Returns a dict of plain floats (JSON-able).  Margin-aware part: `tau` (default 4 x this run's mean |dlogit|; free-running runs
    pass the TEACHER-FORCED run's tau, whose dlogit is arithmetic error only), `flip_rate_by_margin`, `flips_above_margin`;
    `explain=True` (free-running runs; `knn_idx` (G,N,K) + `graph_ids` (B,) 0-based for the neighbourhood rule) adds
    `id_mismatches_explained_frac`."""
    o = [t.detach().float().cpu() if t.dtype != torch.int64 else t.detach().cpu() for t in out]
    r = [t.detach().float().cpu() if t.dtype != torch.int64 else t.detach().cpu() for t in ref]
    zo, zr = torch.cat(o[:3], 1), torch.cat(r[:3], 1)                      # (B, 1+nx+ny, N)
    nx = o[1].shape[1]
    names = ["roi"] + ["x%d" % (nx - 1 - i) for i in range(nx)] + ["y%d" % (o[2].shape[1] - 1 - i) for i in range(o[2].shape[1])]
    agree = (_bit(zo) == _bit(zr)).float().mean(dim=(0, 2))
    d = (zo - zr).abs()
    rms = float(zr.pow(2).mean().sqrt())
    res = {
        "bit_agreement_per_row": {n: round(float(a), 5) for n, a in zip(names, agree)},
        "bit_agreement_min_row": round(float(agree.min()), 5),
        "bit_agreement_all_rows": round(float((_bit(zo) == _bit(zr)).float().mean()), 5),
        "x_id_equal": round(float((o[4] == r[4]).float().mean()), 5),
        "y_id_equal": round(float((o[5] == r[5]).float().mean()), 5),
        "xy_id_equal": round(float(((o[4] == r[4]) & (o[5] == r[5])).float().mean()), 5),
        "id_abs_err_mean_px": round(float(((o[4] - r[4]).abs() + (o[5] - r[5]).abs()).float().mean()), 4),
        "seg_agreement": round(float((_bit(o[3]) == _bit(r[3])).float().mean()), 5),
        "max_abs_dlogit": round(float(d.max()), 5),
        "mean_abs_dlogit": round(float(d.mean()), 6),
        "logit_rms": round(rms, 4),
        "mean_abs_dlogit_over_rms": round(float(d.mean()) / max(rms, 1e-30), 6),
        "max_abs_dseg": round(float((o[3] - r[3]).abs().max()), 5),
    }
    # tau: 4 x this run's mean |dlogit| -- but never more than the ABSOLUTE bound a bf16 pipeline may claim (10 bf16 epsilons of the
    # logit RMS): a build with a larger arithmetic error must not thereby earn a more lenient contract
    res["tau_cap"] = round(TAU_CAP_EPS * BF16_EPS * rms, 6)
    res["tau"] = round(float(tau), 6) if tau is not None else round(min(4.0 * float(d.mean()), res["tau_cap"]), 6)
    res.update(_margin_stats(zo, zr, res["tau"]))
    if explain:
        res.update(_explain_id_mismatches(zo, zr, o, r, res["tau"], knn_idx, graph_ids, init_bits))
    return res


BF16_EPS = 2.0 ** -8                  # relative spacing of bf16 (8 significant bits)
TAU_CAP_EPS = 10.0                    # tau <= 10 eps x logit RMS (3.9 % of the RMS; measured 4 x mean |dlogit| = 3.1 %)
MARGIN_A_FRAC = 0.04                  # clause (a): no flip at a reference margin of max(0.2, 4 % of the logit RMS) or more
MEAN_ERR_CAP_EPS = 2.5                # the mean |dlogit| the margin clauses scale with counts for at most 2.5 eps x logit RMS
MARGIN_BUCKETS = ((0.0, 0.05), (0.05, 0.2), (0.2, 1.0), (1.0, float("inf")))       # of the REFERENCE logit's |z|: its decision margin


def _margin_stats(zo, zr, tau):
    """Where the flipped bits sit relative to the reference's decision margin |z_ref| (a flip at |z| = 2 is an error, a flip at
    |z| = 1e-3 is a coin the reference itself barely decided): flip rate per margin bucket, the largest margin any flip has, and
    the flips whose margin exceeds tau (= 4 x the mean |dlogit| of the run unless given)."""
    flip = _bit(zo) != _bit(zr)
    m = (zr - _Z0).abs()
    buckets = {}
    for lo, hi in MARGIN_BUCKETS:
        sel = (m >= lo) & (m < hi)
        n = int(sel.sum())
        f = int((flip & sel).sum())
        per_row = [(float((flip[:, i] & sel[:, i]).sum()) / max(int(sel[:, i].sum()), 1)) for i in range(zr.shape[1])]
        buckets["%g-%s" % (lo, ("%g" % hi) if hi != float("inf") else "inf")] = {
            "n": n, "flips": f, "flip_rate": round(f / max(n, 1), 6), "worst_row_flip_rate": round(max(per_row), 6)}
    above = flip & (m > tau)
    res = {"flip_rate_by_margin": buckets,
           "flips": int(flip.sum()),
           "flips_above_margin": int(above.sum()),
           "max_flip_margin": round(float(m[flip].max()) if bool(flip.any()) else 0.0, 6),
           "flip_margin_over_dlogit_max": round(float((m[flip] / (zo - zr).abs()[flip].clamp_min(1e-30)).max()) if bool(flip.any()) else 0.0, 4)}
    # the same two tail statistics against the error scale of the flip's OWN logit row / refinement stage (each row is another output
    # head at another depth: pooled, the rows' error scales differ by 2-3x and the pooled mean under-states the deeper heads')
    d = (zo - zr).abs()
    R = zr.shape[1]
    row_mean = d.mean(dim=(0, 2))                                              # (R,)
    if R == 13:
        st = torch.tensor(row_stages(6, 6))
        stage_mean = torch.stack([d[:, st == s_].mean() for s_ in range(int(st.max()) + 1)])[st]     # (R,) the row's stage mean
    else:
        stage_mean = row_mean
    for name, scale in (("row", row_mean), ("stage", stage_mean)):
        rel = m / scale.clamp_min(1e-30)[None, :, None]
        res["max_flip_margin_over_%s_mean" % name] = round(float(rel[flip].max()) if bool(flip.any()) else 0.0, 3)
        res["flips_above_4x_%s_mean" % name] = int((flip & (rel > 4.0)).sum())
    res["row_mean_abs_dlogit"] = [round(float(v), 6) for v in row_mean]
    return res


def row_stages(nx, ny, init_bits=3):
    """refinement stage that PRODUCES each logit row (roi | x bits MSB first | y bits MSB first): InitNet (stage 0) gives roi and the
    `init_bits` most significant bits of each coordinate, every Refine stage one more (pipeline.py:367-381)"""
    return [0] + [max(0, i - (init_bits - 1)) for i in range(nx)] + [max(0, i - (init_bits - 1)) for i in range(ny)]


def _explain_id_mismatches(zo, zr, o, r, tau, knn_idx, graph_ids, init_bits):
    """Free-running runs: the bits of stage s pick the pixels stage s + 1 gathers from (discrete feedback, pipeline.py:367-381), and the
    EdgeConv layers mix a keypoint with its graph neighbours, so ONE near-tie flipped early legitimately changes later logits of
    that keypoint and of its neighbourhood by O(1).  A final (x_id, y_id) mismatch of keypoint n is EXPLAINED when the first stage
    at which n's bits differ has, at n, a flipped bit whose reference margin is below tau (a near-tie: either answer is within the
    arithmetic's error), or when an earlier-stage flip sits within n's 3-hop graph neighbourhood (the three EdgeConv layers of a
    stage: n's input was already perturbed by the feedback).  Unexplained mismatches would be arithmetic errors."""
    B, R, N = zr.shape
    nx, ny = o[1].shape[1], o[2].shape[1]
    st = torch.tensor(row_stages(nx, ny, init_bits))
    flip = _bit(zo) != _bit(zr)                                        # (B, R, N)
    m = (zr - _Z0).abs()
    nst = int(st.max()) + 1
    flip_s = torch.stack([flip[:, st == s_].any(1) for s_ in range(nst)], 1)                      # (B, S, N) any flip at stage s
    sub_s = torch.stack([(flip[:, st == s_] & (m[:, st == s_] < tau)).any(1) for s_ in range(nst)], 1)    # ... with a sub-tau one
    mism = (o[4] != r[4]) | (o[5] != r[5])                            # (B, N)
    if knn_idx is not None:
        idx = torch.as_tensor(knn_idx).long().cpu()
        if idx.dim() == 2:
            idx = idx[None]
        g = (torch.as_tensor(graph_ids).long().cpu() if graph_ids is not None else torch.zeros(B, dtype=torch.long))
        if idx.shape[0] == 1:
            g = torch.zeros(B, dtype=torch.long)
        nb = idx[g]                                                    # (B, N, K)
    else:
        nb = None

    def dilate(mask):                                                  # one hop: n or any of its K neighbours
        if nb is None:
            return mask
        return mask | torch.gather(mask[:, None, :].expand(-1, N, -1), 2, nb).any(2)

    # perturbed[b, n]: keypoint n's INPUT to the current stage already differs legitimately from the reference's -- an EXPLAINED flip of
    # an earlier stage lies within the hops the EdgeConv layers since then have traversed (3 per stage: P_s = dilate^3(P_{s-1} | ok
    # flips of stage s - 1)).  Only explained flips seed it: a flip that is neither a near-tie nor inside a perturbed neighbourhood
    # is an arithmetic error, explains nothing downstream, and every mismatch that starts there stays unexplained.
    perturbed = torch.zeros(B, N, dtype=torch.bool)
    explained = torch.zeros(B, N, dtype=torch.bool)
    decided = torch.zeros(B, N, dtype=torch.bool)
    coverage = []
    for s_ in range(nst):
        ok = flip_s[:, s_] & (sub_s[:, s_] | perturbed)                # this stage's flips that ARE explained
        first = flip_s[:, s_] & ~decided                               # keypoints whose first differing stage is s_
        explained |= first & ok
        decided |= flip_s[:, s_]
        coverage.append(round(float(perturbed.float().mean()), 4))
        reach = perturbed | ok
        for _ in range(3):
            reach = dilate(reach)
        perturbed = reach
    n_m = int(mism.sum())
    n_e = int((mism & explained).sum())
    self_sub = mism & torch.stack([flip_s[:, s_] & sub_s[:, s_] for s_ in range(nst)], 0).any(0)
    return {"id_mismatches": n_m, "id_mismatches_explained": n_e,
            "id_mismatches_explained_frac": round(n_e / n_m, 5) if n_m else 1.0,
            "id_mismatches_from_subtau_self_flip": int(self_sub.sum()),
            # how much of the crop the neighbourhood rule covers when each stage starts: near 1.0 the rule explains any later
            # mismatch (a vacuous pass is visible here), so clause (d) ALSO asks for a share of own near-ties
            "perturbed_coverage_by_stage": coverage,
            "id_mismatches_self_subtau_frac": round(int(self_sub.sum()) / n_m, 5) if n_m else 1.0}


def margin_contract_violations(forced, free=None):
    """The margin-aware part of the bf16 contract, as a list of violated clauses (empty = holds).
    teacher-forced (dlogit = arithmetic error only): (a) no flipped bit whose reference margin |z| is 0.2 -- or 4 % of the logit
    RMS, whichever is larger -- or more; (b) the largest
    margin of any flip <= 6 x the run's mean |dlogit|; (c) flips above tau = 4 x mean |dlogit| are stragglers: <= max(2, 1 %) of
    the flips (a flip needs |dlogit| > margin, so their number follows the tail of the error distribution; measured 0-2).
    free-running (tau taken from the teacher-forced run): (d) >= 95 % of the final id mismatches are explained by an upstream
    near-tie (see _explain_id_mismatches: only explained flips propagate), and -- because the neighbourhood rule covers most of a
    512-keypoint crop after one stage -- >= 60 % of them by the keypoint's OWN sub-tau flip (measured 80-84 %).
    The mean |dlogit| that (b) scales with counts for at most MEAN_ERR_CAP_EPS bf16 epsilons of the logit RMS, and tau is capped
    likewise (logit_agreement): a larger arithmetic error cannot buy a more lenient contract."""
    bad = []
    # (a) is scale-aware: bf16 errors are relative, so a network with larger logits (a trained one: RMS ~10 against ~4.6 at random init)
    # has proportionally larger absolute errors -- the margin no flip may reach is 0.2 or 4 % of the logit RMS, whichever is larger
    thr_a = max(0.2, MARGIN_A_FRAC * forced["logit_rms"])
    if forced["max_flip_margin"] >= thr_a:
        bad.append("teacher-forced: a flip at reference margin %.4f >= %.3f (0.2 or %.0f %% of the logit RMS %.2f)"
                   % (forced["max_flip_margin"], thr_a, 100 * MARGIN_A_FRAC, forced["logit_rms"]))
    mean_err = min(forced["mean_abs_dlogit"], MEAN_ERR_CAP_EPS * BF16_EPS * forced["logit_rms"])
    if forced["mean_abs_dlogit"] > MEAN_ERR_CAP_EPS * BF16_EPS * forced["logit_rms"]:
        bad.append("teacher-forced: mean |dlogit| %.5f > %.1f bf16 epsilons of the logit RMS %.3f" % (forced["mean_abs_dlogit"], MEAN_ERR_CAP_EPS, forced["logit_rms"]))
    if forced["max_flip_margin"] > 6.0 * mean_err:
        bad.append("teacher-forced: a flip at margin %.4f > 6 x mean |dlogit| = %.4f" % (forced["max_flip_margin"], 6 * mean_err))
    if forced["flips_above_margin"] > max(2, 0.01 * forced["flips"]):
        bad.append("teacher-forced: %d of %d flips above tau = %.4f" % (forced["flips_above_margin"], forced["flips"], forced["tau"]))
    if free is not None and "id_mismatches_explained_frac" in free and free["id_mismatches_explained_frac"] < 0.95:
        bad.append("free-running: only %.1f %% of %d id mismatches explained by an upstream sub-tau flip"
                   % (100 * free["id_mismatches_explained_frac"], free["id_mismatches"]))
    if free is not None and free.get("id_mismatches", 0) >= 20 and free.get("id_mismatches_self_subtau_frac", 1.0) < 0.60:
        bad.append("free-running: only %.1f %% of %d id mismatches start at the keypoint's own near-tie (the neighbourhood rule alone "
                   "explains the rest)" % (100 * free["id_mismatches_self_subtau_frac"], free["id_mismatches"]))
    return bad


# ---- attribution of the bf16 error to the three block groups (round 6)
GROUPS = ("backbone", "decoder", "gnn")


def _tail_stats(out, ref, init_bits=3):
    """compact error statistics of one teacher-forced run against the fp32 run: mean / max |dlogit|, the tail (max and the 99.9th
    percentile over the mean), flips by reference margin, and the same mean / max per refinement stage's rows"""
    zo = torch.cat([t.detach().float().cpu() for t in out[:3]], 1)
    zr = torch.cat([t.detach().float().cpu() for t in ref[:3]], 1)
    d = (zo - zr).abs()
    mean = float(d.mean())
    st = torch.tensor(row_stages(out[1].shape[1], out[2].shape[1], init_bits))
    ms = _margin_stats(zo, zr, 4.0 * mean)
    res = {"mean_abs_dlogit": round(mean, 6), "max_abs_dlogit": round(float(d.max()), 5),
           "max_over_mean": round(float(d.max()) / max(mean, 1e-30), 2),
           "p999_over_mean": round(float(torch.quantile(d.flatten()[:: max(1, d.numel() // 4_000_000)], 0.999)) / max(mean, 1e-30), 2),
           "rms_dlogit": round(float(d.pow(2).mean().sqrt()), 6),
           "flips": ms["flips"], "max_flip_margin": ms["max_flip_margin"],
           "max_flip_margin_over_mean": round(ms["max_flip_margin"] / max(mean, 1e-30), 2),
           "flips_by_margin": {k: v["flips"] for k, v in ms["flip_rate_by_margin"].items()},
           "by_stage": {}}
    for s_ in range(int(st.max()) + 1):
        ds = d[:, st == s_]
        res["by_stage"]["stage%d" % s_] = {"mean": round(float(ds.mean()), 6), "max": round(float(ds.max()), 5)}
    return res


def attribute_groups(net, img, obj_ids=None, log=None):
    """Where the bf16 path's logit error comes from: the network is split into `backbone` (image -> the four pyramid features),
    `decoder` (features -> the three maps the refinement stages gather from) and `gnn` (InitNet's head + the three refinement
    stages: everything per keypoint), and each group runs in bf16 ALONE -- the other two in fp32 -- by exchanging the group
    boundaries (NCHW fp32 tensors) between the fp32 and the bf16 programs of the same weights (`forward_hooks`).  Every run is
    teacher-forced with the fp32 logits, so a row's error is arithmetic only.  Returns {group or "all" or "head": _tail_stats}.
    `rss_check` = sqrt(sum of the three groups' rms^2) / rms of the all-bf16 run: ~1 when the groups' errors add independently."""
    say = log or (lambda s: None)
    was = net.compute_dtype
    B, N = img.shape[0], net.npoint
    with torch.no_grad():
        net.set_compute_dtype("fp32")
        ref, side32 = net.forward_hooks(img, want_feats=True, want_dec=True, obj_ids=obj_ids)
        ref = [t.clone() for t in ref]
        feats32 = [t.clone() for t in side32["img_feats"]]
        dec32 = [t.clone() for t in side32["dec_feats"]]
        t = torch.zeros(B, 13, N, device=img.device)
        t[:, 0:1], t[:, 1:1 + ref[1].shape[1]], t[:, 7:7 + ref[2].shape[1]] = ref[0], ref[1], ref[2]
        say("attribution: fp32 reference done")
        net.set_compute_dtype("bf16")
        out_all, side16 = net.forward_hooks(img, teacher_bits=t, want_feats=True, obj_ids=obj_ids)
        out_all = [x.clone() for x in out_all]
        feats16 = [x.clone() for x in side16["img_feats"]]
        out_head, sd = net.forward_hooks(img, teacher_bits=t, feats=feats32, want_dec=True, obj_ids=obj_ids)
        out_head = [x.clone() for x in out_head]
        dec16 = [x.clone() for x in sd["dec_feats"]]
        out_gnn, _ = net.forward_hooks(img, teacher_bits=t, feats=feats32, dec=dec32, obj_ids=obj_ids)
        out_gnn = [x.clone() for x in out_gnn]
        say("attribution: bf16 runs done")
        net.set_compute_dtype("fp32")
        out_bb, _ = net.forward_hooks(img, teacher_bits=t, feats=feats16, obj_ids=obj_ids)
        out_bb = [x.clone() for x in out_bb]
        out_dec, _ = net.forward_hooks(img, teacher_bits=t, feats=feats32, dec=dec16, obj_ids=obj_ids)
        out_dec = [x.clone() for x in out_dec]
        net.set_compute_dtype(was)
    res = {"all": _tail_stats(out_all, ref), "backbone": _tail_stats(out_bb, ref), "decoder": _tail_stats(out_dec, ref),
           "gnn": _tail_stats(out_gnn, ref), "head(decoder+gnn)": _tail_stats(out_head, ref)}
    rss = sum(res[g]["rms_dlogit"] ** 2 for g in GROUPS) ** 0.5
    res["rss_check"] = round(rss / max(res["all"]["rms_dlogit"], 1e-30), 3)
    res["logit_rms"] = round(float(torch.cat(ref[:3], 1).float().pow(2).mean().sqrt()), 4)
    return res
