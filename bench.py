#!/usr/bin/env python3
"""bench.py -- forward throughput of the CheckerPose hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32] [--npoint 512]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one forward of PoseNet_GNNskip (HRNet-W18 + decoder + 3 progressive GNN stages,
hr18GNN2_res6_gnn3Skip_mlpQuery, LM-O `ape` keypoints, npt=512) over one batch of B synthetic 256x256 crops
already resident in HBM; every rank runs its own replica on its own batch shard (pure data parallel: the
forward has no collective), so scaling is weak and value = N*B*K / max-over-ranks(time).
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the
MFMA implicit-GEMM conv) and `cpu_baseline` (oracle restatement timed on the host cores).
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
MFMA_KERNELS = {"conv_igemm": "conv_igemm_kernel", "conv3x3_halo": "conv3x3_halo_kernel", "conv3x3_halo4": "conv3x3_halo4_kernel",
                "conv3x3_halo_s": "conv3x3_halo_s_kernel", "gemm_rows": "gemm_rows_kernel|gemm_rows_ws_kernel",
                "basicblock_fused": "basicblock_fused_kernel|basicblock_persist_kernel", "bottleneck_fused": "bottleneck_fused_kernel"}


def build(npoint, seed=1):
    from tests.common import build_net
    return build_net(npoint=npoint, seed=seed)


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


def cpu_baseline(npoint, seconds=12.0):
    """The oracle (validated CPU restatement incl. its HRNet-W18) on the host cores: B=1 forwards for ~`seconds`."""
    from oracle import checkerpose_oracle as O
    from tests.common import det_image, oracle_kwargs
    torch.set_num_threads(host_threads())
    net = build(npoint)
    sd = net.state_dict()
    img = det_image(1, seed=0)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.posenet_forward(sd, img, net.init_net.knn_idx, npoint, **oracle_kwargs())      # warm-up
        first = time.perf_counter() - t0
        n, t0 = 0, time.perf_counter()
        while (time.perf_counter() - t0 < seconds and first < seconds) or n < 1:          # bounded: ~`seconds` of CPU work
            O.posenet_forward(sd, img, net.init_net.knn_idx, npoint, **oracle_kwargs())
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d forwards at B=1 (fp32, eval, no_grad) of the oracle restatement incl. HRNet-W18, %.1f s" % (n, dt)}


def pmc_traffic_mb(kernel_prefix, dtype, B):
    """HBM bytes per launch (MB) of a kernel family from the committed rocprofv3 PMC summary of this same bench command
    (profiles/r*_<dtype>_b<B>_kernel_summary.csv: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied
    by profiles/summarize.py).  None when no summary for this dtype is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_b%d_kernel_summary.csv" % (dtype, B))))
    if not files:
        return None
    tot, calls = 0.0, 0
    for r in csv.DictReader(open(files[-1])):
        hit = r["kernel"] == kernel_prefix
        if hit and r["avg_hbm_read_MB(FETCH_SIZE*2)"] and r["avg_hbm_write_MB"]:
            n = int(r["calls"])
            tot += n * (float(r["avg_hbm_read_MB(FETCH_SIZE*2)"]) + float(r["avg_hbm_write_MB"]))
            calls += n
    return round(tot / calls, 2) if calls else None


def kernel_breakdown(net, B, steps, dump=None):
    """Per-kernel device time of one step, measured live with HIP events on the launch stream (eager replay of the
    same launch program, one event pair per launch).  Returns (per family, per kernel symbol): the symbol of every
    launch is what the library reports through cp_last_kernel(), i.e. the row name in rocprofv3's kernel stats."""
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
            e0.record(stream)
            fn(sp, *args[1:])
            e1.record(stream)
            if it == 0:
                syms.append(lib.cp_last_kernel().decode())
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
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("CHECKERPOSE_BENCH_BACKEND", "nccl")    # "gloo": exercise the N>1 path on a 1-GPU box
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", (local % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":                                          # nccl == RCCL over xGMI on ROCm
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.set_grad_enabled(False)

    from tests.common import det_image
    net = build(a.npoint).to(dev).set_compute_dtype(a.dtype)
    net.clone_outputs = False            # outputs stay in the program's persistent buffers (no per-step clones)
    B = a.batch
    img = det_image(B, seed=100 + rank).to(dev)      # this rank's shard, resident in HBM before timing

    def step():
        return net(img, None)

    step()                               # builds the launch program for B
    buf = net.input_buffer(B)            # zero-copy boundary: the crops live in the buffer the program reads
    buf.copy_(img)
    img = buf
    for _ in range(max(a.warmup, 2)):    # >= 2: eager run, then hipGraph capture
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        from checkerpose_amd.parallel import max_over_ranks
        el = max_over_ranks(el, dev if backend == "nccl" else None)   # slowest rank defines the whole-job step time
    ms_per_step = el / a.steps * 1e3
    value = world * B * a.steps / el

    out = {"metric": "crops/sec forward (256x256, npt=%d)" % a.npoint, "value": round(value, 1), "unit": "crops/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "LMO 'ape' hr18GNN2_res6_gnn3Skip_mlpQuery npt=%d, PoseNet_GNNskip forward, "
                                  "deterministic random-init weights" % a.npoint,
                      "crops_per_gpu_per_step": B, "global_batch": world * B, "parallelism": "dp%d (no forward collective)" % world,
                      "launch": "hipGraph replay of %d kernel launches" % len(net.program_for(B).calls)}}
    if rank == 0:
        prog = net.program_for(B)
        if not a.no_breakdown:
            fam, symbols = kernel_breakdown(net, B, min(a.steps, 5), a.dump_convs)
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
                          "tflops": round(tf, 1), "frac_mfma": round(tf / PEAK_TFLOPS[a.dtype], 4),
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
                           "frac_mfma": round(tf / PEAK_TFLOPS[a.dtype], 4), "alg_gbs": round(gb, 0),
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
                                   "achieved": pd["tflops"], "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                                   "frac": pd["frac_mfma"], "traffic": None,
                                   "algorithmic_gflop_per_launch_avg": round(sv["flops"] / n / 1e9, 3),
                                   "avg_launch_us": round(pd["ms_per_step"] * 1e3 / n, 2)}
            out["kernel_symbols"] = dict(sorted(ksym.items(), key=lambda kv: -kv[1]["ms_per_step"])[:8])
            if True:         # a committed PMC summary of this exact command (same dtype and batch), if any
                tr = pmc_traffic_mb(dom, a.dtype, B)
                if tr is not None:
                    out["roofline"]["traffic"] = tr
                    out["roofline"]["traffic_unit"] = "MB of HBM read+write per launch (rocprofv3 PMC, profiles/)"
            tot_s = sum(v["ms_per_step"] for v in mf.values()) * 1e-3
            out["mfma_kernels"] = per
            out["mfma_all"] = {"achieved": round(prog.flops / tot_s / 1e12, 2), "unit": "TFLOP/s",
                               "frac": round(prog.flops / tot_s / 1e12 / PEAK_TFLOPS[a.dtype], 4)}
            # memory-bound neighbour-gather kernel: algorithmic bytes = N*K*C*e + 2*N*C*e + N*K*4 per layer
            eg = fam.get("edge_gather")
            if eg:
                e = 2 if a.dtype == "bf16" else 4
                N, K = a.npoint, 20
                by = B * sum(N * K * c * e + 2 * N * c * e + N * K * 4 for c in (64, 64) + (256,) * 9)
                gbs = by / (eg["ms_per_step"] * 1e-3) / 1e9
                hbm_by = B * sum(2 * N * c * e for c in (64, 64) + (256,) * 9)       # compulsory: read P'|Q' once, write out
                out["roofline_gather"] = {"bound": "l2", "kernel": "edgeconv_gather_max_kernel (11 launches)",
                                          "achieved": round(gbs, 1), "peak": 34500.0, "unit": "GB/s",
                                          "frac": round(gbs / 34500.0, 4), "traffic": None,
                                          "algorithmic_mb_per_step": round(by / 1e6, 1),
                                          "compulsory_hbm_gbs": round(hbm_by / (eg["ms_per_step"] * 1e-3) / 1e9, 1),
                                          "note": "algorithmic bytes = K=20 neighbour rows + centre + write + idx (SURVEY.md 8d); the "
                                                  "neighbour rows are served by the XCD L2 (a crop's P' table is 256-512 KiB, all its "
                                                  "blocks share one XCD), so the roof is the ~34.5 TB/s aggregate L2, not HBM"}
            out["kernel_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])}
            out["dense_gflop_per_crop"] = round(prog.flops / B / 1e9, 2)
            out["workspace_mb"] = round(prog.workspace_bytes / 2 ** 20, 1)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.npoint)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
