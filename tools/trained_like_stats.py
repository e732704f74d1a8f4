import sys
sys.path.insert(0, ".")
from checkerpose_amd.trained_like import train_then_measure
for i in range(5):
    r = train_then_measure(npoint=512, steps=300, batch=32, lr=5e-4, held_out=4)
    tf, fr = r["teacher_forced"], r["free_running"]
    print("run %d: loss %s->%s roi_acc %.3f | tf margin %.3f bound %.3f  1-inf flips %d minrow %.4f dl/rms %.4f | fr minrow %.4f ideq %.4f px %.3f expl %.3f viol %s" % (
        i, r["loss_every_25_steps"][0], r["loss_every_25_steps"][-1], r["held_out"]["roi_bit_accuracy_vs_gt"], tf["max_flip_margin"],
        max(0.2, 0.04 * tf["logit_rms"]), tf["flip_rate_by_margin"]["1-inf"]["flips"], tf["bit_agreement_min_row"], tf["mean_abs_dlogit_over_rms"],
        fr["bit_agreement_min_row"], fr["xy_id_equal"], fr["id_abs_err_mean_px"], fr["id_mismatches_explained_frac"], r["margin_contract_violations"]), flush=True)
