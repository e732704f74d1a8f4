#!/usr/bin/env python3
"""bench_train.py -- training-step throughput of the drop-in CheckerPose modules on MI355X (SURVEY.md 8f row N1 / BASELINE
config "DP batch sharded across 8 x MI355X with RCCL grad all-reduce").  NOT the headline metric (that is bench.py, the
forward); same launch contract:

  python bench_train.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench_train.py --gpus N

A "step" = the body of reference train.py:300-320 on one batch of B synthetic 256x256 crops per GPU, already resident in
HBM: zero_grad, PoseNet_GNNskip forward in train mode (batch-statistics BatchNorm), the five losses, backward through the
HIP training program, ONE all-reduce of the flat fp32 gradient buffer over RCCL (N > 1), the Adam step (checkerpose_amd.optim.Adam:
one launch; CHECKERPOSE_BENCH_TORCH_ADAM=1: torch's fused Adam).
Prints one JSON line on rank 0: crops/s (whole job), ms/step, and the per-kernel device time of the step's launch
program (HIP events, eager replay) with achieved TFLOP/s of the weight-gradient kernel.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(npoint, seconds=20.0, B=2):
    """The oracle restatement in train mode (bn_train) + torch autograd on the host cores: forward + backward of B=2 crops
    (the smallest batch with meaningful batch statistics), repeated for ~`seconds`."""
    from bench import host_threads
    from oracle import checkerpose_oracle as O
    from checkerpose_amd.synthetic import build_net, det_image, det_tensor
    from bench import ORACLE_KW
    oracle_kwargs = lambda: ORACLE_KW   # noqa: E731
    torch.set_num_threads(host_threads())
    net = build_net(npoint=npoint, seed=1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    params = [k for k, _ in net.named_parameters()]
    for k in params:
        sd[k].requires_grad_(True)
    img = det_image(B, seed=0)
    seeds = [det_tensor("g_roi", (B, 1, npoint)), det_tensor("g_x", (B, 6, npoint)), det_tensor("g_y", (B, 6, npoint)),
             det_tensor("g_seg", (B, 2, 64, 64), 0.05)]

    def one():
        with torch.enable_grad(), O.bn_train():
            (roi, xb, yb, seg, _, _), _ = O.posenet_forward(sd, img, net.init_net.knn_idx, npoint, **oracle_kwargs())
            torch.autograd.grad([roi, xb, yb, seg], [sd[k] for k in params], seeds, allow_unused=True)

    t0 = time.perf_counter()
    one()
    first = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0 < seconds and first < seconds) or n < 1:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * B / dt, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d forward+backward passes at B=%d (fp32, train-mode BatchNorm, torch autograd over the oracle restatement "
                      "incl. HRNet-W18), %.1f s" % (n, B, dt)}


def run_step_bench(rk, batch=32, npoint=512, dtype="bf16", steps=40, warmup=10, breakdown=True):
    """The training step on this rank's device (rk: bench.Ranks); returns the result dict (complete on rank 0)."""
    from checkerpose_amd.losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
    from checkerpose_amd.losses.mask_loss import MaskLoss_interpolate
    from checkerpose_amd.synthetic import build_net, det_image, det_tensor
    world, rank, dev = rk.world, rk.rank, rk.dev
    B, N = batch, npoint
    net = build_net(npoint=N, seed=1).to(dev).train()
    net.set_compute_dtype(dtype)
    img = det_image(B, seed=100 + rank).to(dev)
    roi_gt = (det_tensor("t_roi", (B, 1, N), seed=rank) > -0.5).float().to(dev)
    x_gt = (det_tensor("t_x", (B, 16, N), seed=rank) > 0).float().to(dev)
    y_gt = (det_tensor("t_y", (B, 16, N), seed=rank) > 0).float().to(dev)
    m_vis = (det_tensor("t_mv", (B, 128, 128), seed=rank) > 0).float().to(dev)
    m_full = (det_tensor("t_mf", (B, 128, 128), seed=rank) > -0.3).float().to(dev)
    roi_loss, bit_loss, seg_loss = UnmaskedCodeLoss("BCE"), MaskedCodeLoss("BCE"), MaskLoss_interpolate()
    from checkerpose_amd.optim import Adam                 # optim.Adam(net.parameters(), lr) of train.py:246 as ONE launch per step
    opt = (torch.optim.Adam(net.parameters(), lr=2e-4, fused=True) if os.environ.get("CHECKERPOSE_BENCH_TORCH_ADAM") == "1"
           else Adam(net.parameters(), lr=2e-4))
    p3d = torch.zeros(1, 3, N, device=dev).expand(B, -1, -1)
    last = [None, None]

    def step():
        opt.zero_grad(set_to_none=True)
        roi, xb, yb, seg, _, _ = net(img, p3d, 3)
        nb = xb.shape[1]
        loss = roi_loss(roi, roi_gt) + bit_loss(xb, x_gt[:, :nb], roi_gt) + bit_loss(yb, y_gt[:, :nb], roi_gt) \
            + seg_loss(seg[:, 0:1], m_vis) + seg_loss(seg[:, 1:2], m_full)
        loss.backward()
        opt.step()
        last[1] = loss
        return loss

    for _ in range(max(warmup, 1)):
        l0 = step()
    el, seen, per_rank = rk.timed(step, steps, torch.cuda.synchronize)
    l1 = last[1]
    pr = list(net._train_programs.values())[0]
    prog = pr["prog"]
    out = {"metric": "crops/sec training step (256x256, npt=%d)" % N, "value": round(world * B * steps / el, 1), "unit": "crops/s",
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
           "config": {"workload": "LMO 'ape' hr18GNN2_res6_gnn3Skip_mlpQuery npt=%d: train.py step (forward in train mode, 5 losses, "
                                  "backward, gradient all-reduce, Adam)" % N,
                      "crops_per_gpu_per_step": B, "global_batch": world * B,
                      "parallelism": "dp%d, one all-reduce of the %.1f MB flat fp32 gradient buffer per step" % (world, pr["pgrad"].numel() * 4 / 1e6),
                      "launches": "%d forward + %d backward kernel launches per step" % (prog.n_fwd_ops, len(prog.calls) - prog.n_fwd_ops)},
           "ranks_seen": seen, "per_rank_ms": per_rank, "backend": rk.backend if world > 1 else None,
           "loss_first_last": [round(float(l0), 4), round(float(l1), 4)],
           "workspace_mb": round(prog.workspace_bytes / 2 ** 20, 1)}
    if rank == 0 and breakdown:
        from checkerpose_amd import _abi
        lib = _abi.load()
        stream = torch.cuda.current_stream()
        sp = stream.cuda_stream
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in prog.calls]
        syms = []
        for (fn, args, name), (e0, e1) in zip(prog.calls, evs):
            lib.cp_kernel_log_begin()
            e0.record(stream)
            fn(sp, *args[1:])
            e1.record(stream)
            # every symbol the call launched ("a + b" for a call with several launches: it is timed, and named, as the set)
            syms.append((lib.cp_kernel_log().decode() or name.split(":")[0])
                        if not name.startswith(("memset", "grad_zero", "pgrad_zero", "refresh_vec", "save_ids", "memcpy")) else name.split(":")[0])
        torch.cuda.synchronize()
        agg = {}
        for i, ((fn, args, name), (e0, e1), sy) in enumerate(zip(prog.calls, evs, syms)):
            half = "fwd" if i < prog.n_fwd_ops else "bwd"
            r = agg.setdefault((half, sy), [0.0, 0])
            r[0] += e0.elapsed_time(e1)
            r[1] += 1
        per_call = {}
        for (fn, args, name), (e0, e1) in zip(prog.calls, evs):
            if name.startswith(("wgrad", "bn_bwd", "bn_stats")):
                r = per_call.setdefault(name, [0.0, 0])
                r[0] += e0.elapsed_time(e1)
                r[1] += 1
        out["slowest_named_calls"] = {k: {"ms": round(v[0], 3), "n": v[1]} for k, v in sorted(per_call.items(), key=lambda kv: -kv[1][0])[:16]}
        rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
        out["kernel_ms_per_step"] = {"%s:%s" % k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in rows[:24]}
        out["device_ms_fwd_bwd"] = [round(sum(v[0] for k, v in agg.items() if k[0] == h), 3) for h in ("fwd", "bwd")]
        # roofline of the dominant dense training kernel: the all-taps weight gradient (MFMA-bound), algorithmic FLOPs
        # 2*M*9*Cin*Cout of its launches / their measured device time (HIP events on the launch stream)
        peak = {"bf16": 2500.0, "fp32": 157.3}[dtype]
        w3 = [(i, e0.elapsed_time(e1)) for i, ((fn, args, name), (e0, e1), sy) in enumerate(zip(prog.calls, evs, syms))
              if sy.startswith("wgrad") and i in prog.wgrad_flops and " k3 s1 " in name]
        if w3:
            ms = sum(t for _, t in w3)
            fl = sum(prog.wgrad_flops[i] for i, _ in w3)
            big = max(w3, key=lambda it: prog.wgrad_flops[it[0]])
            from bench import _profile_prefix_mb_per_step
            n_single = sum(1 for i, _ in w3 if not prog.calls[i][2].startswith("wgrad_group"))
            tr, trf = _profile_prefix_mb_per_step("wgrad", "train_%s_b%d" % (dtype, B))
            out["roofline"] = {"bound": "mfma", "kernel": "all-taps 3x3 weight gradient: wgrad3x3_kernel (%d single launches per step, the layers above %g GFLOP) + "
                                                          "wgrad_group_kernel (%d grouped launches: the smaller layers, several per launch); the pixel-slice partials "
                                                          "are summed by the batched wgrad_reduce launches" % (n_single, prog.wg_group_flops / 1e9, len(w3) - n_single),
                               "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4),
                               "traffic": tr, "traffic_unit": "MB of HBM read+write per STEP, all weight-gradient kernels incl. the reductions (rocprofv3 PMC, %s)" % trf,
                               "algorithmic_gflop_per_step": round(fl / 1e9, 1), "ms_per_step": round(ms, 3),
                               "largest_launch": {"name": prog.calls[big[0]][2], "gflop": round(prog.wgrad_flops[big[0]] / 1e9, 1),
                                                  "us": round(big[1] * 1e3, 1),
                                                  "tflops": round(prog.wgrad_flops[big[0]] / (big[1] * 1e-3) / 1e12, 1)}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps: the first ~10 replays of freshly captured hipGraphs run slower (one-time, ~100 ms in total)")
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU per step (reference config: batch_size 32)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--npoint", type=int, default=512)
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--deterministic", action="store_true", help="time the deterministic training mode (cp_set_deterministic) as the main line")
    ap.add_argument("--no-deterministic-cost", action="store_true", help="skip the second, deterministic-mode measurement of the default run")
    a = ap.parse_args()
    from bench import Ranks, self_launch
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # not under torchrun: launch the N ranks ourselves (no GPU call here)
        sys.exit(self_launch(__file__, a.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit("bench_train: --gpus %d but the launcher started %s ranks" % (a.gpus, os.environ["WORLD_SIZE"]))
    rk = Ranks()
    import checkerpose_amd
    checkerpose_amd.set_deterministic(a.deterministic)
    out = run_step_bench(rk, a.batch, a.npoint, a.dtype, a.steps, a.warmup, not a.no_breakdown)
    out["deterministic"] = bool(a.deterministic)
    if rk.world == 1 and not a.deterministic and not a.no_deterministic_cost:
        # what bit-reproducible training costs: the same step with every accumulation in a fixed order (include/checkerpose_hip.h)
        checkerpose_amd.set_deterministic(True)
        try:
            d = run_step_bench(rk, a.batch, a.npoint, a.dtype, max(a.steps // 2, 5), 5, False)
            out["deterministic_mode"] = {"crops_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                                         "slowdown": round(d["ms_per_step"] / out["ms_per_step"], 3),
                                         "note": "CHECKERPOSE_AMD_DETERMINISTIC=1 / checkerpose_amd.set_deterministic(True): BatchNorm sums one block per "
                                                 "accumulator set, weight gradients through the fixed-order reduction, Index2Feat scatter as an ordered gather"}
        finally:
            checkerpose_amd.set_deterministic(False)
    if rk.rank == 0 and rk.world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.npoint)
    if rk.rank == 0:
        print(json.dumps(out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
