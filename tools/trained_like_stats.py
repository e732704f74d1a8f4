#!/usr/bin/env python3
"""The bf16 contract on trained-like weights over several seeds (GPU box): deterministic training mode, `--seeds` runs of `--steps`
steps each, then bf16 (keypoint side in half, the bench's kernel selection) vs fp32 on held-out crops.  One line per run + a JSON
list (gpurun_out/trained_like_stats.json).
  python tools/trained_like_stats.py [--seeds 1,2,3,4,5] [--steps 300] [--held-out 8]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1,2,3,4,5")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--held-out", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/trained_like_stats.json")
    a = ap.parse_args()
    import checkerpose_amd
    from checkerpose_amd.trained_like import train_then_measure
    checkerpose_amd.set_deterministic(True)
    res = []
    for seed in [int(s) for s in a.seeds.split(",")]:
        r = train_then_measure(npoint=512, steps=a.steps, batch=32, lr=5e-4, seed=seed, held_out=a.held_out)
        r["seed"] = seed
        res.append(r)
        tf, fr = r["teacher_forced"], r["free_running"]
        print("seed %d steps %d: loss %s->%s roi_acc %.3f | tf mean %.5f (%.4f of rms %.2f) max %.4f flips %d max flip margin %.4f (%.2f x mean) "
              "above tau %d | per row: %.2f x, %d above 4x | per stage: %.2f x, %d above 4x | minrow %.4f | fr minrow %.4f ideq %.4f px %.3f expl %.3f self %.3f | viol %s" % (
                  seed, a.steps, r["loss_every_25_steps"][0], r["loss_every_25_steps"][-1], r["held_out"]["roi_bit_accuracy_vs_gt"],
                  tf["mean_abs_dlogit"], tf["mean_abs_dlogit_over_rms"], tf["logit_rms"], tf["max_abs_dlogit"], tf["flips"], tf["max_flip_margin"],
                  tf["max_flip_margin"] / max(tf["mean_abs_dlogit"], 1e-30), tf["flips_above_margin"], tf["max_flip_margin_over_row_mean"], tf["flips_above_4x_row_mean"],
                  tf["max_flip_margin_over_stage_mean"], tf["flips_above_4x_stage_mean"], tf["bit_agreement_min_row"],
                  fr["bit_agreement_min_row"], fr["xy_id_equal"], fr["id_abs_err_mean_px"], fr["id_mismatches_explained_frac"],
                  fr.get("id_mismatches_self_subtau_frac", 1.0), r["margin_contract_violations"]), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"))
    sys.exit(1 if any(r["margin_contract_violations"] for r in res) else 0)          # the dedicated check: a violation fails the call
