#!/usr/bin/env python3
"""bench.py -- forward throughput of the CheckerPose hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32] [--npoint 512]
                  [--workload lmo_ape|ycbv_rr21|lm13_n4096] [--no-extras] [--dry-run]
  N>1: either way works -- under the driver's launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
       --master-addr 127.0.0.1 ... bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or plain
       `python bench.py --gpus N`: with no WORLD_SIZE in the environment the process makes no GPU call, starts exactly that
       launcher as a child and exits with its code.  The line carries `ranks_seen` (all-gather of the rank ids) and `per_rank_ms`.

A "step" = one forward of PoseNet_GNNskip (HRNet-W18 + decoder + 3 progressive GNN stages,
hr18GNN2_res6_gnn3Skip_mlpQuery, LM-O `ape` keypoints, npt=512) over one batch of B synthetic 256x256 crops
already resident in HBM; every rank runs its own replica on its own batch shard (pure data parallel: the
forward has no collective), so scaling is weak and value = N*B*K / max-over-ranks(time).
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the
MFMA implicit-GEMM conv) and `cpu_baseline` (oracle restatement timed on the host cores).  Extra keys of the default
N=1 run (`--no-extras` skips them): `fp32_exact` (the path north_star's 1e-4 applies to, at its best
batch), `by_batch` (bf16 at B = 1 / 8 / 32 / 64 / 128: the reference's test.py:198 runs batch 1, its configs train at 32),
`b1_latency_ms`, `bf16_agreement` (the bf16 path's accuracy contract, checkerpose_amd/agreement.py, measured against the
fp32 HIP path on the same crops) and `host_u8` (crops start as uint8 in pinned HOST memory and are double-buffered over
PCIe on a copy stream: the PCIe-inclusive rate, never `value`) `device_crop` (the data loader's RoI crop + resize done on the
device from full frames in HBM, row N3), `end_to_end` (frames + boxes -> poses without leaving the GPU) and `configs`: short
sub-runs of BASELINE configs #4 (`ycbv_rr21`), #5 (`lm13_n4096`) and the training step (`train_step_b32`, bench_train.py), each with
its own value / ms_per_step / batch / roofline.

Workloads (BASELINE.json configs):
  lmo_ape     (default) config #2: LM-O `ape`, one network, npt=512
  ycbv_rr21   config #4: the reference trains ONE network per YCB-V object (train.py:384,396) -> 21 independent
              (weights, kNN graph) sets, step i runs model i mod 21 on its own batch
  lm13_n4096  config #5: the LM shared estimator (pipeline_lm.py:392-425) at npt=4096, obj_ids uniform over the 13 LM ids
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
LM13_DEFAULT_BATCH = 256      # config #5 (`--workload lm13_n4096`): crops per GPU per step; measured 7 010 / 9 010 / 10 290 / 10 940 crops/s at 32 / 64 / 128 / 256
MFMA_KERNELS = {"conv_igemm": "conv_igemm_kernel", "conv3x3_halo": "conv3x3_halo_kernel", "conv3x3_halo4": "conv3x3_halo4_kernel",
                "conv3x3_halo_s": "conv3x3_halo_s_kernel", "conv3x3_s2_small": "conv3x3_s2_small_kernel", "gemm_rows": "gemm_rows_kernel|gemm_rows_ws_kernel",
                "basicblock_fused": "basicblock_fused_kernel|basicblock_persist_kernel", "bottleneck_fused": "bottleneck_fused_kernel",
                "hr_chain": "hr_chain_kernel|hr_chain0_kernel", "hr_fuse_out": "hr_fuse_out_kernel", "edge_fused": "edgeconv_fused_kernel", "edge_tiled": "edgeconv_ptable_kernel + edgeconv_tiled_kernel", "hr_stem": "hr_stem_kernel", "patch_gather": "patch_gather_kernel", "mlp_fused": "mlp_query_fused_kernel|mlp_pair_fused_kernel"}


ORACLE_KW = dict(backbone="hrnet_w18", res_log2=6, init_n_graph=2, n_graph=3, local_k=2, slope=0.01, graph_slope=0.2,
                 init_graph_slope=0.2)     # hr18GNN2_res6_gnn3Skip_mlpQuery.txt:14-28


def build(npoint, seed=1, **kw):
    from checkerpose_amd.synthetic import build_net
    return build_net(npoint=npoint, seed=seed, **kw)


def host_threads():
    """Threads the CPU baseline may really use: affinity mask and cgroup CPU quota, not os.cpu_count() (the GPU box
    reports 256 logical CPUs; oversubscribing torch's intra-op pool on them is pathologically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(npoint, seconds=9.0):
    """The oracle (validated CPU restatement incl. its HRNet-W18) on the host cores: B=1 forwards under no_grad for ~`seconds`
    (`value`), then the two variants SURVEY.md 8(d) names -- the reference's own test mode (test.py:290 never enters no_grad, so
    autograd records the graph) and B=8 -- one bounded sample each (~20 s of CPU work in total)."""
    from oracle import checkerpose_oracle as O
    from checkerpose_amd.synthetic import det_image
    oracle_kwargs = lambda: ORACLE_KW   # noqa: E731
    torch.set_num_threads(host_threads())
    net = build(npoint)
    sd = net.state_dict()
    img = det_image(1, seed=0)

    def sample(fn, budget, first_counts=False):
        t0 = time.perf_counter()
        fn()                                                            # warm-up (counted when one call already exhausts the budget)
        first = time.perf_counter() - t0
        if first_counts or first >= budget:
            return 1, first
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget or n < 1:               # bounded: ~`budget` seconds of CPU work
            fn()
            n += 1
        return n, time.perf_counter() - t0

    with torch.no_grad():
        n, dt = sample(lambda: O.posenet_forward(sd, img, net.init_net.knn_idx, npoint, **oracle_kwargs()), seconds)
    res = {"value": round(n / dt, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d forwards at B=1 (fp32, eval, no_grad) of the oracle restatement incl. HRNet-W18, %.1f s" % (n, dt)}
    try:
        sd_g = dict(sd)
        sd_g.update({k: p_ for k, p_ in net.named_parameters()})         # parameters that require grad: the graph is recorded
        with torch.enable_grad():
            n2, dt2 = sample(lambda: O.posenet_forward(sd_g, img, net.init_net.knn_idx, npoint, **oracle_kwargs()), 4.0)
        img8 = det_image(8, seed=0)
        with torch.no_grad():
            n8, dt8 = sample(lambda: O.posenet_forward(sd, img8, net.init_net.knn_idx, npoint, **oracle_kwargs()), 4.0, first_counts=True)
        res["variants"] = {"b1_autograd_recording": {"value": round(n2 / dt2, 3), "sample": "%d forwards at B=1 WITHOUT no_grad (the reference's "
                                                     "test.py:290 mode), %.1f s" % (n2, dt2)},
                           "b8_no_grad": {"value": round(8 * n8 / dt8, 3), "sample": "%d forward(s) at B=8 under no_grad, %.1f s" % (n8, dt8)}}
    except Exception as e:
        res["variants"] = {"error": repr(e)[:200]}
    return res


def profile_tag(workload, dtype, B):
    return ("%s_b%d" % (dtype, B)) if workload == "lmo_ape" else ("%s_%s_b%d" % (workload, dtype, B))


def _summary_files(tag):
    """profiles/rNN_<tag>_kernel_summary.csv, oldest round first -- the tag must follow the round prefix directly (a plain glob on
    `r*_bf16_b256_*` also matched the `ycbv_rr21_bf16_b256` / `lm13_n4096_bf16_b256` summaries and the default line quoted theirs)"""
    import glob
    import re
    pat = re.compile(r"^r\d+[a-z]?_%s_kernel_summary\.csv$" % re.escape(tag))
    return sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_summary.csv")) if pat.match(os.path.basename(f)))


def committed_profile(kernel_prefix, tag):
    """What the committed rocprofv3 evidence of this same bench command says about one kernel symbol
    (profiles/r*_<tag>_kernel_summary.csv, made by profiles/summarize.py from a `--kernel-trace --stats` run and two separate
    `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes with the gfx950 corrections): HBM MB per launch, rocprofv3's average launch
    duration in us, and the file.  (None, None, None) when no summary for this workload / dtype / batch is committed.  These
    figures are NOT re-measured by this run: they go stale when the kernel changes, hence the file name beside them.
    `kernel_prefix` may name a SET ("a<..> + b<..>": one entry point, several launches): the durations and the traffic of its
    parts are SUMMED (a part missing from the summary voids the figure)."""
    import csv
    files = _summary_files(tag)
    if not files:
        return None, None, None
    rows = list(csv.DictReader(open(files[-1])))
    tr_sum, us_sum = 0.0, 0.0
    for part in [p.strip() for p in kernel_prefix.split(" + ")]:
        tot, calls, dur, dcalls = 0.0, 0, 0.0, 0
        for r in rows:
            if r["kernel"] != part:
                continue
            n = int(r["calls"])
            dur += n * float(r["avg_us"])
            dcalls += n
            if r["avg_hbm_read_MB(FETCH_SIZE*2)"] and r["avg_hbm_write_MB"]:
                tot += n * (float(r["avg_hbm_read_MB(FETCH_SIZE*2)"]) + float(r["avg_hbm_write_MB"]))
                calls += n
        tr_sum = (tr_sum + tot / calls) if (calls and tr_sum is not None) else None
        us_sum = (us_sum + dur / dcalls) if (dcalls and us_sum is not None) else None
    return ((round(tr_sum, 2) if tr_sum else None), (round(us_sum, 2) if us_sum else None), os.path.relpath(files[-1], ROOT))


def _profile_prefix_mb_per_step(prefix, tag):
    """sum over the rows of the committed summary whose kernel name starts with `prefix` (template instances): HBM MB per STEP"""
    import csv
    import glob
    files = _summary_files(tag)
    if not files:
        return None, None
    rows = [r for r in csv.DictReader(open(files[-1])) if r["kernel"].startswith(prefix) and r["avg_hbm_read_MB(FETCH_SIZE*2)"]]
    if not rows:
        return None, None
    steps = _profile_steps(files[-1])
    mb = sum(int(r["calls"]) * (float(r["avg_hbm_read_MB(FETCH_SIZE*2)"]) + float(r["avg_hbm_write_MB"])) for r in rows)
    return round(mb / steps, 1), os.path.relpath(files[-1], ROOT)


def _profile_steps(path):
    """forwards covered by a committed kernel summary: the stats run is `bench.py --steps 5` = 5 timed + 3 warm-up + 1 program
    build forward = 9 (every decoder launch appears exactly twice per forward: calls of the dominant kernel / 2)"""
    import csv
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        if r["kernel"].startswith("conv3x3_halo4_kernel<BF16Tag, false, true"):
            return max(int(r["calls"]) // 2, 1)
    for r in rows:                                  # a bench_train.py summary: three code losses (roi, x bits, y bits) per training step
        if r["kernel"] == "code_loss_kernel":
            return max(int(r["calls"]) // 3, 1)
    return 9


def timed_steps(step, steps, warmup):
    """`warmup` untimed calls (>= 2: eager run, then hipGraph capture), then `steps` timed ones; seconds."""
    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def side_measurements(net_bf16, npoint, dev, B_main, img_main):
    """The numbers the headline value does not carry (all on this GPU, synthetic crops resident in HBM unless stated)."""
    from checkerpose_amd.agreement import logit_agreement
    from checkerpose_amd.synthetic import det_image, det_tensor
    ex = {}
    # ---- bf16 at the small batches the reference itself uses (test.py:198 batch 1; config batch 32)
    by = {}
    for b in (1, 8, 32, 64, 128):     # 64 / 128: the per-crop launches (from 40 crops), low-resolution chains one / two crops per workgroup
        img = det_image(b, seed=200 + b).to(dev)
        net_bf16(img, None)
        buf = net_bf16.input_buffer(b); buf.copy_(img)
        n = 200 if b == 1 else 50
        el = timed_steps(lambda: net_bf16(buf, None), n, 3)
        by[str(b)] = {"crops_per_s": round(b * n / el, 1), "ms_per_step": round(el / n * 1e3, 3)}
    ex["by_batch"] = {"dtype": "bf16", "note": "hipGraph replay, crops resident in HBM", **by}
    ex["b1_latency_ms"] = by["1"]["ms_per_step"]
    # ---- the fp32-exact path (logits within 1e-4 of the CPU reference: tests/test_gpu_parity.py) at its best batch
    net32 = build(npoint).to(dev).set_compute_dtype("fp32")
    net32.clone_outputs = False
    b32 = 128
    img = det_image(b32, seed=300).to(dev)
    net32(img, None)
    buf = net32.input_buffer(b32); buf.copy_(img)
    el = timed_steps(lambda: net32(buf, None), 8, 2)
    ex["fp32_exact"] = {"crops_per_s": round(b32 * 8 / el, 1), "ms_per_step": round(el / 8 * 1e3, 3), "batch": b32,
                        "dtype": "fp32", "note": "exact-fp32 MFMA path; the only path north_star's 1e-4 logit criterion applies to"}
    # ---- accuracy contract of the timed bf16 path, against the fp32 HIP path on the same crops (B=8)
    net32.clone_outputs = True
    img8 = det_image(8, seed=3).to(dev)
    ref = net32(img8, None)
    # measured on the launches the TIMED step is made of: the per-crop kernel selection (what "auto" picks at 256 crops: LDS-resident
    # chains / EdgeConv / MLP stacks, keypoint side in IEEE half) pinned for this 8-crop batch -- "auto" would pick the small-batch kernels
    net_pc = build(npoint).to(dev).set_compute_dtype("bf16")
    net_pc.load_state_dict(net_bf16.state_dict())
    net_pc.set_kernel_selection("per_crop")
    net_pc.clone_outputs = True
    t = torch.zeros(8, 13, npoint, device=dev)
    t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
    forced = logit_agreement(net_pc.forward_teacher_forced(img8, t), ref)
    free = logit_agreement(net_pc(img8, None), ref, tau=forced["tau"], explain=True, knn_idx=net_pc.init_net.knn_idx)
    half_on = bool(net_pc.program_for(8).progs[0].gnn_half)
    del net_pc
    from checkerpose_amd.agreement import margin_contract_violations
    keep = ("bit_agreement_min_row", "bit_agreement_all_rows", "xy_id_equal", "id_abs_err_mean_px", "seg_agreement",
            "max_abs_dlogit", "mean_abs_dlogit", "logit_rms", "tau", "flips", "flips_above_margin", "max_flip_margin",
            "flip_rate_by_margin", "id_mismatches", "id_mismatches_explained_frac", "id_mismatches_from_subtau_self_flip",
            "id_mismatches_self_subtau_frac", "perturbed_coverage_by_stage", "tau_cap")
    ex["bf16_agreement"] = {"vs": "fp32 HIP path (oracle-pinned <= 1e-4) on 8 crops, random-init weights",
                            "kernel_selection": "per_crop (the launches of the timed step)", "keypoint_side_half": half_on,
                            "margin_contract_violations": margin_contract_violations(forced, free),
                            "free_running": {k: free[k] for k in keep if k in free}, "free_running_rows": free["bit_agreement_per_row"],
                            "teacher_forced": {k: forced[k] for k in keep if k in forced}, "teacher_forced_rows": forced["bit_agreement_per_row"]}
    # ---- the same contract on TRAINED-LIKE weights: 300 steps of the HIP training program on a synthetic task whose targets are a
    #      function of the image (checkerpose_amd/trained_like.py), then bf16 vs fp32 eval on held-out crops (~15 s)
    #      Deterministic training mode: the trained network is the same in every run.  `attribution`: each block group (backbone |
    #      decoder | gnn) in bf16 ALONE against the fp32 path.  The contract itself is ASSERTED in tests/test_gpu_train_step.py and
    #      checked by `tools/trained_like_stats.py` (non-zero exit on a violation); here it is reported.
    try:
        import checkerpose_amd
        from checkerpose_amd.trained_like import train_then_measure
        checkerpose_amd.set_deterministic(True)
        try:
            ex["bf16_agreement"]["trained_like"] = train_then_measure(npoint, device=dev, attribution=(npoint == 512))
        finally:
            checkerpose_amd.set_deterministic(False)
    except Exception as e:          # an extra must never take the headline down
        ex["bf16_agreement"]["trained_like"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    # ---- post-forward rows N2 + N4 on the device: correspondences + EPnP / RANSAC pose for the whole batch (opt-in path; the
    #      reference does both per image on the host, test_network_with_test_data.py:32-115).  Random-init weights: the poses are
    #      meaningless, the launch time is what is measured.
    try:
        import numpy as np
        from checkerpose_amd.postprocess import correspondences, solve_pnp_ransac
        from checkerpose_amd.synthetic import DATA
        out_main = net_bf16(img_main, None)
        grid = (det_tensor("roi_xy", (B_main, 2, 64, 64), 300.0) + 320.0).to(dev)
        xyz = torch.from_numpy(np.load(os.path.join(DATA, "fps_lmo_obj01.npy"))[:npoint]).float().to(dev)
        Kc = torch.tensor([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]], device=dev)

        def post():
            p2d, valid, _ = correspondences(out_main, grid)
            return solve_pnp_ransac(xyz, p2d, valid, Kc)
        el = timed_steps(post, 10, 2)
        st_ = post()[3]
        ex["post_forward"] = {"ms_per_batch": round(el / 10 * 1e3, 3), "batch": B_main, "solved": int(st_.sum()),
                              "note": "cp_correspondences + cp_pnp_ransac (150 EPnP hypotheses per crop, fp64) for the whole batch, outputs "
                                      "never leave the device; opt-in rows N2 + N4, not part of `value`"}
    except Exception as e:          # an extra must never take the headline down
        ex["post_forward"] = {"error": repr(e)[:200]}
    del net32
    torch.cuda.empty_cache()
    # ---- PCIe-inclusive feed: uint8 HWC crops in pinned host memory -> H2D on a copy stream (double-buffered) ->
    #      cp_u8hwc_to_nhwc_norm + forward on the compute stream (the stated 8-GPU limiter)
    B = B_main
    host = [(det_tensor("u8feed%d" % i, (B, 256, 256, 3)).abs() * 255.999).to(torch.uint8).pin_memory() for i in range(2)]
    stage = [torch.empty(B, 256, 256, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
    copy_s = torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    for i in range(2):
        done[i].record(cur)

    def feed_step(i):
        k = i & 1
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(done[k])                      # the forward that last read stage[k] has finished
            stage[k].copy_(host[k], non_blocking=True)
            ready[k].record(copy_s)
        cur.wait_event(ready[k])
        net_bf16(stage[k], None)
        done[k].record(cur)

    for i in range(4):
        feed_step(i)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for i in range(n):
        feed_step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ex["host_u8"] = {"crops_per_s": round(B * n / el, 1), "ms_per_step": round(el / n * 1e3, 3), "batch": B,
                     "h2d_gb_per_s": round(B * n * 256 * 256 * 3 / el / 1e9, 2),
                     "note": "PCIe-inclusive: uint8 crops from pinned host memory, H2D double-buffered on a copy stream, "
                             "normalised on the device (cp_u8hwc_to_nhwc_norm); not `value`"}
    # ---- the loader's crop on the device (row N3, second half): 32 full 640 x 480 uint8 frames resident in HBM, 8 detections per
    #      frame -> padding_Bbox / crop_square_resize / cv2-style bilinear resize in ONE launch (cp_crop_resize_u8) -> uint8 forward
    from checkerpose_amd import preprocess as PP
    g = torch.Generator().manual_seed(11)
    frames = torch.randint(0, 256, (32, 480, 640, 3), dtype=torch.uint8, generator=g).to(dev)
    boxes = [PP.padding_Bbox([int(torch.randint(0, 560, (1,), generator=g)), int(torch.randint(0, 400, (1,), generator=g)),
                              int(torch.randint(40, 200, (1,), generator=g)), int(torch.randint(40, 200, (1,), generator=g))], 1.5)
             for _ in range(B)]
    fidx = [b % 32 for b in range(B)]
    crops = torch.empty(B, 256, 256, 3, dtype=torch.uint8, device=dev)

    def crop_step():
        PP.get_roi_batch(frames, boxes, 256, PP.INTER_LINEAR, "crop_square_resize", img_index=fidx, out=crops)
        net_bf16(crops, None)
    el = timed_steps(crop_step, 10, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        PP.get_roi_batch(frames, boxes, 256, PP.INTER_LINEAR, "crop_square_resize", img_index=fidx, out=crops)
    e1.record()
    torch.cuda.synchronize()
    ex["device_crop"] = {"crops_per_s": round(B * 10 / el, 1), "ms_per_step": round(el / 10 * 1e3, 3), "batch": B,
                         "crop_kernel_ms": round(e0.elapsed_time(e1) / 10, 3),
                         "note": "full uint8 frames in HBM -> RoI windows + 8-bit bilinear resize on the device (one launch per batch, "
                                 "incl. the host-side window arithmetic and its 6 KB upload) -> uint8 forward; not `value`"}
    # ---- test.py's inner loop end to end on the device: detection boxes on frames in HBM -> crops -> forward -> correspondences from
    #      the final boxes -> EPnP + RANSAC; 12 doubles + a status word per crop leave the GPU (rows N3 + path + N2 + N4)
    try:
        from checkerpose_amd import postprocess as Q
        raw_boxes = [[int(b[0] + b[2] // 6), int(b[1] + b[3] // 6), max(int(b[2] * 2 // 3), 8), max(int(b[3] * 2 // 3), 8)] for b in boxes]
        p3d_xyz = det_tensor("e2e_p3d", (npoint, 3), 60.0).to(dev)
        K = torch.tensor([[572.4, 0.0, 325.3], [0.0, 573.6, 242.0], [0.0, 0.0, 1.0]])

        def e2e_step():
            return Q.estimate_poses(net_bf16, frames, raw_boxes, p3d_xyz, K, img_index=fidx)
        el = timed_steps(e2e_step, 10, 3)
        st_ = e2e_step()[3]
        ex["end_to_end"] = {"crops_per_s": round(B * 10 / el, 1), "ms_per_step": round(el / 10 * 1e3, 3), "batch": B, "solved": int(st_.sum()),
                            "note": "frames in HBM + detection boxes -> poses (crop, forward, correspondences, EPnP + RANSAC), random-init "
                                    "weights: the correspondences are noise, RANSAC runs its full 150 hypotheses; not `value`"}
    except Exception as e:                                   # an extra must never take the headline line down
        ex["end_to_end"] = {"error": repr(e)[:200]}
    return ex


def kernel_breakdown(net, B, steps, dump=None):
    """Per-kernel device time of one step, measured live with HIP events on the launch stream (eager replay of the
    same launch program, one event pair per launch).  Returns (per family, per kernel symbol): the symbol of every
    launch is what the library reports through cp_kernel_log(), i.e. the row name(s) in rocprofv3's kernel stats; an entry
    point that issues several launches (cp_edgeconv_tiled: key table + gather) is ONE row named "a<..> + b<..>"."""
    from checkerpose_amd import _abi
    lib = _abi.load()
    prog = net.program_for(B)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    fam, sym = {}, {}
    syms = []
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in prog.calls]
    for it in range(steps):
        for (fn, args, name), (e0, e1) in zip(prog.calls, evs):
            if it == 0:
                lib.cp_kernel_log_begin()
            e0.record(stream)
            fn(sp, *args[1:])
            e1.record(stream)
            if it == 0:          # an entry point that issues SEVERAL launches is priced as the set: "a<..> + b<..>", launch order
                syms.append(lib.cp_kernel_log().decode() or lib.cp_last_kernel().decode())
        torch.cuda.synchronize()
        for (fn, args, name), (e0, e1), sy in zip(prog.calls, evs, syms):
            k = name.split(":")[0]
            ms = e0.elapsed_time(e1)
            t, c = fam.get(k, (0.0, 0))
            fam[k] = (t + ms, c + 1)
            r = sym.setdefault(sy, {"ms": 0.0, "n": 0, "flops": 0, "bytes": 0, "family": k})
            r["ms"] += ms
            r["n"] += 1
    per_conv = []
    ci = 0
    conv_log = prog.conv_log
    for (fn, args, name), (e0, e1), sy in zip(prog.calls, evs, syms):
        if name.split(":")[0] in MFMA_KERNELS:
            wk, M, Cout, K, fl, kfam, nby = conv_log[ci]
            ci += 1
            assert kfam == name.split(":")[0], "conv_log out of step with the launch list: %s vs %s" % (kfam, name)
            sym[sy]["flops"] += fl
            sym[sy]["bytes"] += nby
            ms = e0.elapsed_time(e1)          # last step's duration of this launch
            per_conv.append({"name": wk, "kernel": sy, "M": M, "Cout": Cout, "K": K, "gflop": round(fl / 1e9, 3), "us": round(ms * 1e3, 1),
                             "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0,
                             "alg_gbs": round(nby / (ms * 1e-3) / 1e9, 0) if ms > 0 else 0})
    if dump:
        with open(dump, "w") as f:
            json.dump(sorted(per_conv, key=lambda r: -r["us"]), f, indent=0)
    fams = {k: {"ms_per_step": t / steps, "launches_per_step": c // steps} for k, (t, c) in fam.items()}
    symbols = {k: {"ms_per_step": r["ms"] / steps, "launches_per_step": r["n"] // steps, "flops": r["flops"], "bytes": r["bytes"],
                   "family": r["family"]} for k, r in sym.items()}
    return fams, symbols


def self_launch(script, n):
    """`python <script> --gpus N` outside a torchrun environment: this process makes NO GPU call; it starts a fresh child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... <script> <same arguments>` (a subprocess, never an exec:
    the pool forbids replacing a process that may have touched the GPU), lets rank 0's JSON line through on the shared stdout and
    exits with the child's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def gpu_local_cpus(index, sysfs="/sys/class/drm"):
    """the host cores next to the `index`-th AMD GPU of this node: `local_cpulist` of its PCI device (sysfs; display-class AMD
    devices in PCI bus order, the order ROCm enumerates them in) -> sorted core list, or None when sysfs does not say.
    Pure file reads: no GPU call."""
    import glob
    devs = {}
    for card in glob.glob(os.path.join(sysfs, "card[0-9]*")):
        d = os.path.join(card, "device")
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                continue
            if not open(os.path.join(d, "class")).read().strip().startswith(("0x03", "0x12")):      # display / processing accelerator
                continue
            devs[os.path.basename(os.path.realpath(d))] = open(os.path.join(d, "local_cpulist")).read().strip()
        except OSError:
            continue
    if index >= len(devs):
        return None
    cpus = set()
    for part in devs[sorted(devs)[index]].split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus) or None


def pin_rank_to_gpu_numa_node(local_rank, world):
    """N > 1: bind this rank's host threads to the cores of its GPU's NUMA node BEFORE the first GPU call (the PCIe-inclusive legs --
    `host_u8`, the data-parallel step's host side -- then feed each GPU from local memory).  Returns what was done, for the line."""
    if world <= 1 or os.environ.get("CHECKERPOSE_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = gpu_local_cpus(local_rank)
        allowed = os.sched_getaffinity(0)
        cpus = sorted(set(cpus or ()) & allowed)
        if not cpus or len(cpus) == len(allowed):
            return None
        os.sched_setaffinity(0, cpus)
        return "%d cores: %d-%d" % (len(cpus), cpus[0], cpus[-1])
    except (OSError, ValueError):
        return None


class Ranks:
    """This process's place in the job: torchrun's environment, the process group, the device."""

    def __init__(self, dry_run=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("CHECKERPOSE_BENCH_BACKEND", "nccl")    # "gloo": exercise the N>1 path on a 1-GPU box / on CPU
        self.dist, self.dev = None, None
        self.cpu_affinity = pin_rank_to_gpu_numa_node(self.local, self.world) if self.backend == "nccl" else None   # before any GPU call
        if not dry_run:
            ndev = torch.cuda.device_count()
            if self.world > 1 and self.backend == "nccl" and ndev < self.world:
                raise SystemExit("bench: %d ranks over RCCL need %d GPUs, this node shows %d (CHECKERPOSE_BENCH_BACKEND=gloo rehearses "
                                 "the N>1 path with several ranks per GPU)" % (self.world, self.world, ndev))
            self.dev = torch.device("cuda", (self.local % max(ndev, 1)) if self.world > 1 else 0)
            torch.cuda.set_device(self.dev)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl" and not dry_run:                        # nccl == RCCL over xGMI on ROCm
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo" if dry_run else self.backend)
            self.dist = dist
        self.coll_dev = self.dev if (self.backend == "nccl" and not dry_run) else None     # where the tiny timing collectives live

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def timed(self, step, steps, sync):
        """barrier + sync, EXACTLY `steps` steps, sync + barrier; returns (whole-job seconds = MAX over ranks, ranks seen, per-rank ms)"""
        from checkerpose_amd.parallel import gather_over_ranks, max_over_ranks
        sync()
        self.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        el_local = time.perf_counter() - t0
        self.barrier()
        el = max_over_ranks(el_local, self.coll_dev)                           # slowest rank defines the whole-job step time
        seen = [int(r) for r in gather_over_ranks(self.rank, self.coll_dev)]
        per_rank = [round(v / steps * 1e3, 3) for v in gather_over_ranks(el_local, self.coll_dev)]
        return el, seen, per_rank

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def make_workload(workload, npoint, B, dtype, dev, rank=0, streams=1, backbone="hrnet_w18"):
    """Networks + resident inputs + the step closure of one BASELINE config.  Returns dict(net, step, name, img, nets)."""
    from checkerpose_amd.synthetic import LM_OBJ_IDS, det_image, ycbv_p3d
    img = det_image(B, seed=100 + rank).to(dev)      # this rank's shard, resident in HBM before timing
    if workload == "ycbv_rr21":
        # 21 per-object networks (own weights seed, own kNN graph); every model gets its program + graph before timing
        nets = [build(npoint, seed=o, p3d=ycbv_p3d(o, npoint)).to(dev).set_compute_dtype(dtype) for o in range(1, 22)]
        for n_ in nets:
            n_.clone_outputs = False
            for _ in range(2):
                n_(img, None)
        bufs = []
        for n_ in nets:
            b_ = n_.input_buffer(B); b_.copy_(img); bufs.append(b_)
        counter = [0]
        # the 21 per-object networks are independent (own weights, graph, workspace, hipGraph): with --streams S > 1 consecutive
        # steps go to S HIP streams in turn.  Measured on MI355X (B = 64): 14 960 / 15 700 / 15 330 / 15 430 crops/s at S = 1 / 2 / 4 /
        # 6 -- ROCm does not overlap the replays of independent multi-lane hipGraphs, so the default stays 1
        rr_streams = [torch.cuda.Stream(dev) for _ in range(max(streams, 1))] if streams > 1 else None
        if rr_streams:
            for s_ in rr_streams:
                s_.wait_stream(torch.cuda.current_stream(dev))

        def step():
            k = counter[0] % 21
            counter[0] += 1
            if rr_streams is None:
                return nets[k](bufs[k], None)
            with torch.cuda.stream(rr_streams[counter[0] % len(rr_streams)]):
                return nets[k](bufs[k], None)
        name = "YCB-V all 21 objects, one hr18GNN2_res6_gnn3Skip_mlpQuery network per object (round-robin), npt=%d" % npoint
        return {"net": nets[0], "nets": nets, "step": step, "name": name, "img": bufs[0]}
    lm = workload == "lm13_n4096"
    net = build(npoint, lm=lm, backbone=backbone).to(dev).set_compute_dtype(dtype)
    net.clone_outputs = False            # outputs stay in the program's persistent buffers (no per-step clones)
    obj = torch.tensor([LM_OBJ_IDS[(i + rank) % 13] for i in range(B)], device=dev) if lm else None
    net(img, None, obj) if lm else net(img, None)      # builds the launch program for B
    buf = net.input_buffer(B)            # zero-copy boundary: the crops live in the buffer the program reads
    buf.copy_(img)

    def step():
        return net(buf, None, obj) if lm else net(buf, None)
    name = ("LM 13-object shared estimator (pipeline_lm), npt=%d dense keypoints, obj_ids uniform over the 13 LM ids" % npoint
            if lm else "LMO 'ape' hr18GNN2_res6_gnn3Skip_mlpQuery npt=%d" % npoint)
    if backbone != "hrnet_w18":
        name += " with backbone %s (NOT BASELINE's config)" % backbone
    return {"net": net, "nets": [net], "step": step, "name": name, "img": buf}


def roofline_report(out, net, B, dtype, workload, npoint, steps, dump=None):
    """`roofline` (+ `roofline_gather`, `kernel_symbols`, `mfma_kernels`, ...) of one configuration into `out`: live HIP-event
    timing of every launch (kernel_breakdown) priced with the launch program's algorithmic FLOPs / bytes."""
    prog = net.program_for(B)
    fam, symbols = kernel_breakdown(net, B, steps, dump)
    # MFMA kernel families (host-side grouping) ...
    fl_by, by_by = {}, {}
    for wk, M, Cout, K, fl, kf, nby in prog.conv_log:
        fl_by[kf] = fl_by.get(kf, 0) + fl
        by_by[kf] = by_by.get(kf, 0) + nby
    mf = {k: fam[k] for k in MFMA_KERNELS if k in fam}
    per = {}
    for k, v in mf.items():
        t = v["ms_per_step"] * 1e-3
        tf, gb = fl_by.get(k, 0) / t / 1e12, by_by.get(k, 0) / t / 1e9
        per[k] = {"kernel": MFMA_KERNELS[k], "ms_per_step": round(v["ms_per_step"], 3), "launches": v["launches_per_step"],
                  "tflops": round(tf, 1), "frac_mfma": round(tf / PEAK_TFLOPS[dtype], 4),
                  "alg_gbs": round(gb, 0), "frac_hbm": round(gb / PEAK_HBM_GBS, 4)}
    # ... and per kernel SYMBOL (the granularity of rocprofv3's kernel stats): `roofline` = the symbol with the
    # most device time per step, priced against the roof it sits closer to
    ksym = {}
    for k, v in symbols.items():
        if not v["flops"]:
            continue
        t = v["ms_per_step"] * 1e-3
        tf, gb = v["flops"] / t / 1e12, v["bytes"] / t / 1e9
        ksym[k] = {"ms_per_step": round(v["ms_per_step"], 3), "launches": v["launches_per_step"], "tflops": round(tf, 1),
                   "frac_mfma": round(tf / PEAK_TFLOPS[dtype], 4), "alg_gbs": round(gb, 0),
                   "frac_hbm": round(gb / PEAK_HBM_GBS, 4)}
    dom = max(ksym, key=lambda k: ksym[k]["ms_per_step"])
    pd, sv = ksym[dom], symbols[dom]
    n = pd["launches"]
    if pd["frac_hbm"] > pd["frac_mfma"]:
        out["roofline"] = {"bound": "hbm", "kernel": "%s (%d launches per step)" % (dom, n),
                           "achieved": pd["alg_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": pd["frac_hbm"],
                           "traffic": None, "algorithmic_mb_per_launch_avg": round(sv["bytes"] / n / 1e6, 2),
                           "avg_launch_us": round(pd["ms_per_step"] * 1e3 / n, 2)}
    else:
        out["roofline"] = {"bound": "mfma", "kernel": "%s (%d launches per step)" % (dom, n),
                           "achieved": pd["tflops"], "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                           "frac": pd["frac_mfma"], "traffic": None,
                           "algorithmic_gflop_per_launch_avg": round(sv["flops"] / n / 1e9, 3),
                           "avg_launch_us": round(pd["ms_per_step"] * 1e3 / n, 2)}
    out["kernel_symbols"] = dict(sorted(ksym.items(), key=lambda kv: -kv[1]["ms_per_step"])[:8])
    out["roofline"]["timing"] = ("HIP events on the launch stream around every launch of an eager replay of the same launch "
                                 "program, mean over %d steps (bench.py:kernel_breakdown)" % steps)
    short = " + ".join(p_.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70] for p_ in dom.split(" + "))
    tr, us_prof, src = committed_profile(short, profile_tag(workload, dtype, B))
    if src is not None:      # committed rocprofv3 evidence of this exact command (same workload, dtype and batch), if any
        out["roofline"]["traffic"] = tr
        out["roofline"]["traffic_unit"] = "MB of HBM read+write per launch" + (" SET (sum over its %d kernels)" % len(dom.split(" + ")) if " + " in dom else "")
        if tr and sv["bytes"]:                              # moved bytes / algorithmic bytes: > 1 = re-reads, partial-sector stores
            out["roofline"]["traffic_ratio"] = round(tr * 1e6 * n / sv["bytes"], 3)
        out["roofline"]["traffic_source"] = ("from_committed_profile: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                             "this command; not re-measured by this run)" % src)
        if us_prof:          # the same fraction priced with rocprofv3's own average duration of that kernel
            per_launch = (sv["bytes"] / n / 1e3 / us_prof / PEAK_HBM_GBS) if out["roofline"]["bound"] == "hbm" else \
                         (sv["flops"] / n / 1e6 / us_prof / PEAK_TFLOPS[dtype])
            out["roofline"]["rocprofv3_avg_launch_us"] = us_prof
            out["roofline"]["frac_rocprofv3"] = round(per_launch, 4)
    tot_s = sum(v["ms_per_step"] for v in mf.values()) * 1e-3
    out["mfma_kernels"] = per
    out["mfma_all"] = {"achieved": round(prog.flops / tot_s / 1e12, 2), "unit": "TFLOP/s",
                       "frac": round(prog.flops / tot_s / 1e12 / PEAK_TFLOPS[dtype], 4)}
    # the GNN gather (north_star: "achieved GB/s on the GNN gather"): algorithmic bytes per layer =
    # N*K*C*e neighbour rows + N*C*e centre + N*C*e write + N*K*4 index, all 11 EdgeConv layers.  Three kernels can carry
    # it: edgeconv_fused (N = 512: table in LDS), edgeconv_tiled (N = 4096: neighbour window in LDS), and the L2 gather.
    e = 2 if dtype == "bf16" else 4
    N, K = npoint, 20
    LDS_PEAK = 256 * 256 * 2.4          # B/clk/CU x CUs x GHz = GB/s (MI355X_MICROARCH.md: ds_read_b128 256 B/clk/CU)
    layers = {}                          # family -> [C' per launch]
    for (fn_, args_, name_) in prog.calls:
        f_ = name_.split(":")[0]
        if f_ in ("edge_fused", "edge_tiled"):
            layers.setdefault(f_, []).append(64 if name_.split(":")[1].startswith("init_net.") else 256)
        elif f_ == "edge_gather":
            layers.setdefault(f_, []).append(int(name_.split(":")[1]))
    rg = {}
    for f_, cs in layers.items():
        t_ = fam[f_]["ms_per_step"] * 1e-3
        by = B * sum(N * K * c * e + 2 * N * c * e + N * K * 4 for c in cs)
        hbm_by = B * sum(2 * N * c * e for c in cs)            # compulsory: the layer's input rows in, its output out
        lds = f_ != "edge_gather"
        peak = LDS_PEAK if lds else 34500.0
        sym_ = {"edge_fused": "edgeconv_fused_kernel", "edge_tiled": "edgeconv_tiled_kernel", "edge_gather": "edgeconv_gather_max_kernel"}[f_]
        shown = "edgeconv_ptable_kernel + edgeconv_tiled2_kernel (key table + LDS-staged gather launch)" if f_ == "edge_tiled" else sym_
        r_ = {"bound": "lds" if lds else "l2", "kernel": "%s (%d launches per step)" % (shown, len(cs)),
              "achieved": round(by / t_ / 1e9, 1), "peak": round(peak, 0), "unit": "GB/s", "frac": round(by / t_ / 1e9 / peak, 4),
              "algorithmic_mb_per_step": round(by / 1e6, 1), "ms_per_step": round(t_ * 1e3, 3),
              "compulsory_hbm_mb_per_step": round(hbm_by / 1e6, 1), "compulsory_hbm_gbs": round(hbm_by / t_ / 1e9, 1),
              "traffic": None,
              "note": ("gather bytes = K=20 neighbour rows + centre + write + idx (SURVEY.md 8d) over the WHOLE launch time; "
                       + ("the launch also runs the layer's node GEMM on the MFMA pipe, so this is a lower bound of the gather "
                          "rate; the neighbour rows are read from the LDS table (ds_read_b128: %.0f TB/s aggregate)" % (LDS_PEAK / 1e3)
                          if lds else "the neighbour rows are served by the XCD L2 (~34.5 TB/s aggregate), not HBM"))}
        # template instances: by prefix (the tiled family = key-table launch + gather launch: edgeconv_ptable_kernel, edgeconv_tiled2_kernel)
        tr, src = _profile_prefix_mb_per_step("edgeconv_" if f_ == "edge_tiled" else sym_, profile_tag(workload, dtype, B))
        if tr is not None:
            r_["traffic"] = tr
            r_["traffic_unit"] = "MB of HBM read+write per step, all launches of this kernel (PMC)"
            r_["traffic_source"] = "from_committed_profile: %s" % src
            r_["compulsory_hbm_fraction"] = round(hbm_by / 1e6 / tr, 3)
        rg[f_] = r_
    if rg:
        main_f = max(rg, key=lambda k_: rg[k_]["ms_per_step"])
        out["roofline_gather"] = rg[main_f]
        for k_, v_ in rg.items():
            if k_ != main_f:
                out["roofline_gather_" + k_] = v_
    out["kernel_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])}
    out["dense_gflop_per_crop"] = round(prog.flops / B / 1e9, 2)
    out["workspace_mb"] = round(prog.workspace_bytes / 2 ** 20, 1)


# the other BASELINE configs inside the DEFAULT line (N = 1): short runs of the same code paths `--workload` / bench_train.py time
SUB_CONFIGS = (("ycbv_rr21", dict(workload="ycbv_rr21", npoint=512, batch=256, steps=42)),
               ("lm13_n4096", dict(workload="lm13_n4096", npoint=4096, batch=LM13_DEFAULT_BATCH, steps=20)))


def sub_configs(dtype, dev):
    """configs #4 / #5 and the training step (row N1) as short sub-runs of the default line: each its own try / except (an extra
    can never take the headline down), own `roofline` / `roofline_gather`, everything freed before the next."""
    res = {}
    for key, c in SUB_CONFIGS:
        t_start = time.perf_counter()
        try:
            wl = make_workload(c["workload"], c["npoint"], c["batch"], dtype, dev)
            el = timed_steps(wl["step"], c["steps"], 3)
            r = {"metric": "crops/sec forward (256x256, npt=%d)" % c["npoint"], "value": round(c["batch"] * c["steps"] / el, 1), "unit": "crops/s",
                 "ms_per_step": round(el / c["steps"] * 1e3, 3), "batch": c["batch"], "steps": c["steps"], "dtype": dtype,
                 "workload": wl["name"], "command": "bench.py --workload %s --batch %d" % (c["workload"], c["batch"])}
            roofline_report(r, wl["net"], c["batch"], dtype, c["workload"], c["npoint"], 2)
            for k in ("mfma_kernels", "kernel_symbols"):
                r.pop(k, None)                              # the full tables are in the `--workload` line; keep the sub-run compact
            r["kernel_ms_per_step"] = dict(list(r["kernel_ms_per_step"].items())[:8])
            del wl
        except Exception as e:
            r = {"error": repr(e)[:300]}
        r["wall_s"] = round(time.perf_counter() - t_start, 1)
        res[key] = r
        torch.cuda.empty_cache()
    t_start = time.perf_counter()
    try:
        import bench_train
        torch.set_grad_enabled(True)
        r = bench_train.run_step_bench(Ranks_single(dev), batch=32, npoint=512, dtype=dtype, steps=20, warmup=10, breakdown=True)
        r = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "loss_first_last", "device_ms_fwd_bwd", "roofline") if k in r}
        r["batch"], r["steps"], r["command"] = 32, 20, "bench_train.py --batch 32"
    except Exception as e:
        r = {"error": repr(e)[:300]}
    finally:
        torch.set_grad_enabled(False)
    r["wall_s"] = round(time.perf_counter() - t_start, 1)
    res["train_step_b32"] = r
    torch.cuda.empty_cache()
    return res


class Ranks_single:
    """the one-rank stand-in for `Ranks` inside a process that already owns its device (sub-runs of the default line)"""
    world, rank, local, dist, backend, coll_dev = 1, 0, 0, None, "nccl", None

    def __init__(self, dev):
        self.dev = dev

    barrier = lambda self: None          # noqa: E731
    timed = Ranks.timed
    close = lambda self: None            # noqa: E731


def dry_run(a):
    """`--dry-run`: the launcher + rendezvous + timing collectives of the N-rank path with NO GPU call (CPU test of `--gpus N`):
    a step is a 1 ms sleep."""
    rk = Ranks(dry_run=True)
    el, seen, per_rank = rk.timed(lambda: time.sleep(0.001), a.steps, lambda: None)
    if rk.rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": round(rk.world * a.batch * a.steps / el, 1), "unit": "crops/s",
                          "n_gpus": rk.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3),
                          "ranks_seen": seen, "per_rank_ms": per_rank, "dry_run": True, "backend": "gloo"}), flush=True)
    rk.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="crops per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--npoint", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--dump-convs", default=None, help="write per-conv-launch timings (json) to this path")
    ap.add_argument("--workload", default="lmo_ape", choices=["lmo_ape", "ycbv_rr21", "lm13_n4096"])
    ap.add_argument("--backbone", default="hrnet_w18", choices=["hrnet_w18", "hrnet_w18_small", "hrnet_w30", "resnet34"],
                    help="lmo_ape only: another of the reference's backbone names (not BASELINE's config; --no-extras --no-breakdown implied)")
    ap.add_argument("--no-extras", action="store_true", help="skip fp32_exact / by_batch / bf16_agreement / host_u8 / the sub-configs")
    ap.add_argument("--streams", type=int, default=1, help="ycbv_rr21: HIP streams the independent per-object networks' steps rotate over")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / timing collectives only, no GPU call (CPU test of --gpus N)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # not under torchrun: launch the N ranks ourselves
        sys.exit(self_launch(__file__, a.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started %s ranks" % (a.gpus, os.environ["WORLD_SIZE"]))
    if a.dry_run:
        return dry_run(a)
    if a.workload == "lm13_n4096":
        a.npoint = 4096
        if "--batch" not in sys.argv:
            a.batch = LM13_DEFAULT_BATCH
    # (ycbv_rr21 runs at the default 256 crops per network too: switching among the 21 networks costs nothing -- 22 290 crops/s
    #  against 22 470-23 300 for the single network; 14 880 at --batch 64 and 19 540 at 128: the per-crop launches need the crops)

    rk = Ranks()
    world, rank, dev = rk.world, rk.rank, rk.dev
    torch.set_grad_enabled(False)
    B = a.batch
    if a.backbone != "hrnet_w18":
        assert a.workload == "lmo_ape", "--backbone applies to the lmo_ape workload"
        a.no_extras = a.no_breakdown = True
    wl = make_workload(a.workload, a.npoint, B, a.dtype, dev, rank, a.streams, a.backbone)
    net, step, img = wl["net"], wl["step"], wl["img"]
    for _ in range(max(a.warmup, 2)):    # >= 2: eager run, then hipGraph capture
        step()
    el, seen, per_rank = rk.timed(step, a.steps, torch.cuda.synchronize)
    ms_per_step = el / a.steps * 1e3
    value = world * B * a.steps / el

    out = {"metric": "crops/sec forward (256x256, npt=%d)" % a.npoint, "value": round(value, 1), "unit": "crops/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": wl["name"] + ", PoseNet_GNNskip forward, deterministic random-init weights",
                      "crops_per_gpu_per_step": B, "global_batch": world * B, "parallelism": "dp%d (no forward collective)" % world,
                      "launch": "hipGraph replay of %d kernel launches" % len(net.program_for(B).calls),
                      "precision": ("bf16 storage + MFMA inputs on the image side (backbone, decoder), IEEE half (f16 storage + f16 MFMA, same bytes and "
                                    "rate) on the per-keypoint side (EdgeConv, Index2Feat rows, MLP stacks); fp32 accumulation and epilogues"
                                    if (a.dtype == "bf16" and net.program_for(B).progs[0].gnn_half) else a.dtype)},
           "ranks_seen": seen, "per_rank_ms": per_rank, "backend": rk.backend if world > 1 else None,
           "cpu_affinity": rk.cpu_affinity}
    if a.workload == "ycbv_rr21":
        out["config"]["streams"] = max(a.streams, 1)
    if rank == 0:
        if not a.no_breakdown:
            roofline_report(out, net, B, a.dtype, a.workload, a.npoint, min(a.steps, 5), a.dump_convs)
        if world == 1 and not a.no_extras and a.workload == "lmo_ape" and a.dtype == "bf16":
            out.update(side_measurements(net, a.npoint, dev, B, img))
            del wl, net, step, img
            torch.cuda.empty_cache()
            out["configs"] = sub_configs(a.dtype, dev)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.npoint)
        print(json.dumps(out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
