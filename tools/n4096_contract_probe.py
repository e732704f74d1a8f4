import sys, torch
sys.path.insert(0, ".")
from checkerpose_amd import engine
from checkerpose_amd.agreement import logit_agreement, margin_contract_violations
from tests.test_gpu_parity import _lm4096_golden_case, _lm4096_case, _teacher_bits
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for case in ("golden", "oracle"):
  for tiled in (False, True):
    for half in (True, False):
        engine.USE_GNN_F16 = half
        engine.EDGE_FUSED_MIN_BATCH = 1 if tiled else 1 << 30
        engine.MLP_FUSED_MIN_ROWS = 1 if tiled else 1 << 30
        if case == "golden":
            obj, net, feats, ref = _lm4096_golden_case()
            net = net.to(dev).set_compute_dtype("bf16")
            img = torch.zeros(2, 3, 256, 256, device=dev)
            fd = [f.to(dev) for f in feats]
            tf = logit_agreement(net.forward_injected_feats(img, fd, obj_ids=obj.to(dev), teacher_bits=_teacher_bits(ref).to(dev)), ref)
            fr = logit_agreement(net.forward_injected_feats(img, fd, obj_ids=obj.to(dev)), ref, tau=tf["tau"], explain=True, knn_idx=net.init_net.knn_idx, graph_ids=obj - 1)
        else:
            obj, net, img, ref = _lm4096_case()
            net = net.to(dev).set_compute_dtype("bf16")
            tf = logit_agreement(net.forward_teacher_forced(img.to(dev), _teacher_bits(ref).to(dev), obj_ids=obj.to(dev)), ref)
            fr = logit_agreement(net(img.to(dev), None, obj.to(dev)), ref, tau=tf["tau"], explain=True, knn_idx=net.init_net.knn_idx, graph_ids=obj - 1)
        print(case, "tiled", tiled, "half", half, "| tf mean %.5f rms %.3f flips %d maxflip %.4f (%.1fx) above_tau %d minrow %.4f | fr minrow %.4f ideq %.4f expl %.3f self %.3f n_mism %d | %s" % (
            tf["mean_abs_dlogit"], tf["logit_rms"], tf["flips"], tf["max_flip_margin"], tf["max_flip_margin"] / tf["mean_abs_dlogit"], tf["flips_above_margin"], tf["bit_agreement_min_row"],
            fr["bit_agreement_min_row"], fr["xy_id_equal"], fr["id_mismatches_explained_frac"], fr["id_mismatches_self_subtau_frac"], fr["id_mismatches"], margin_contract_violations(tf, fr)), flush=True)
