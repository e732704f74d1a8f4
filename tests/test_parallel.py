"""CPU: the N>1 path (sharding + timing aggregation + output gather) under gloo, world_size 2."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from checkerpose_amd.parallel import (aggregate_crops_per_sec, allreduce_gradients_, gather_outputs, max_over_ranks,
                                      shard_bounds)


def test_shard_bounds_cover_batch_exactly():
    for gb in (0, 1, 7, 32, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(7, rank, world)                 # ragged: 4 + 3 crops
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1) * torch.ones(1, 3)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                         # rank 1 is the slow one
    el = time.perf_counter() - t0
    mx = max_over_ranks(el)
    thr = aggregate_crops_per_sec(hi - lo, el)
    full = gather_outputs(local)
    # gradient all-reduce: three tensors, tiny bucket size -> 2 buckets; mean over ranks
    grads = [torch.full((5,), float(rank + 1)), torch.full((3, 2), 10.0 * (rank + 1)), torch.full((4,), -1.0 * rank)]
    nb = allreduce_gradients_(grads, bucket_bytes=40)
    q.put((rank, lo, hi, el, mx, thr, full.numpy().tolist(), nb, [g.flatten().tolist() for g in grads]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, lo0, hi0, el0, mx0, thr0, full0, nb0, g0), (r1, lo1, hi1, el1, mx1, thr1, full1, nb1, g1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)
    assert mx0 == mx1 and abs(mx0 - max(el0, el1)) < 1e-9          # slowest rank defines the step time
    assert abs(thr0 - 7 / mx0) < 1e-6 and thr0 == thr1             # whole-job crops / max time
    assert full0 == full1 == [[float(i)] * 3 for i in range(7)]    # rank-ordered ragged gather
    assert nb0 == nb1 == 2 and g0 == g1 == [[1.5] * 5, [15.0] * 6, [-0.5] * 4]     # bucketed mean all-reduce


def test_plan_gradient_buckets_covers_buffer_and_orders_segments():
    from checkerpose_amd.parallel import plan_gradient_buckets
    # module order: backbone (largest, final LAST in the backward), init head, decoder, refinement, seg (final FIRST)
    slots = {"bb.a": 0, "bb.b": 400, "init.c": 1000, "up.d": 1100, "ref.e": 1700, "seg.f": 1990, "unused.g": 1996}
    done = {"seg.f": 12, "ref.e": 30, "up.d": 45, "init.c": 60, "bb.b": 80, "bb.a": 95}
    segs = plan_gradient_buckets(slots, 2000, done, first_call=10, n_calls=100, nbuckets=4)
    assert segs[0][0] == 10 and segs[-1][1] == 100
    assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))                       # consecutive launch segments
    spans = sorted((a, b) for _, _, a, b in segs)
    assert spans[0][0] == 0 and spans[-1][1] == 2000 and all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
    for lo, hi, a, b in segs:                                                       # a bucket is reduced only once it is final
        for k, off in slots.items():
            if a <= off < b:
                assert done.get(k, 10) <= hi
    assert segs[0][2] >= 1000 and segs[-1][2] == 0                                  # tail first, backbone last
    assert plan_gradient_buckets(slots, 2000, done, 10, 100, nbuckets=1) == [(10, 100, 0, 2000)]


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from checkerpose_amd.parallel import backward_with_bucketed_allreduce_, plan_gradient_buckets
    slots = {"bb": 0, "head": 300, "seg": 650}
    segs = plan_gradient_buckets(slots, 1000, {"seg": 3, "head": 5, "bb": 9}, first_call=2, n_calls=10, nbuckets=3)
    flat = torch.zeros(1000)
    log = []

    def run_segment(k, lo, hi):                    # the "backward launches" [lo, hi): bucket k's slice becomes final
        a, b = segs[k][2], segs[k][3]
        assert float(flat[a:b].abs().max()) == 0.0                                 # not reduced / written before its segment
        flat[a:b] = float(rank + 1) * (k + 1)
        log.append((k, lo, hi))

    n = backward_with_bucketed_allreduce_(flat, segs, run_segment)
    q.put((rank, segs, n, log, [float(flat[a]) for _, _, a, _ in segs], float(flat.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bucketed_backward_allreduce():
    """the data-parallel backward path of _TrainFn.backward (model/_runtime.py) under gloo: segments run in order, one
    async all-reduce per bucket right behind its segment, mean over ranks in place, identical on both ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, segs0, n0, log0, vals0, sum0), (_, segs1, n1, log1, vals1, sum1) = res
    assert segs0 == segs1 and n0 == n1 == 3 and log0 == log1 == [(k, lo, hi) for k, (lo, hi, _, _) in enumerate(segs0)]
    assert vals0 == vals1 == [1.5 * (k + 1) for k in range(3)]                      # mean of (1, 2) * (k + 1)
    assert sum0 == sum1


def test_bench_gpus_n_self_launches_n_ranks_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts `torch.distributed.run --nproc-per-node 2` itself,
    rank 0's line comes through and says who took part (no GPU call anywhere: --dry-run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CHECKERPOSE_BENCH_BACKEND"] = "gloo"
    for script in ("bench.py",):           # bench_train.py uses the same launcher (bench.self_launch); its ranks need a GPU
        p = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout                       # ONE JSON line, from rank 0
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and len(out["per_rank_ms"]) == 2
        assert out["steps"] == 5 and out["dry_run"] is True
        assert out["ms_per_step"] >= max(out["per_rank_ms"]) - 1e-6        # whole-job time = the slowest rank's


def test_bench_rejects_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=120, cwd=root)
    assert p.returncode != 0 and "launcher started 2 ranks" in (p.stderr + p.stdout)


def test_dp_child_plan_picks_rccl_when_the_node_has_several_gpus():
    """tests/conftest.py: the GPU-side data-parallel job (tests/dp_step_child.py + `bench.py --gpus N`) runs one rank per GPU over
    RCCL as soon as the box shows two or more (config #3's gradient all-reduce over xGMI: never exercised on the 1-GPU boxes), up to 4
    (six GPU processes per job on this pool; the 8-rank case is the driver's); a 1-GPU box keeps the two-rank gloo rehearsal"""
    from tests.conftest import dp_child_plan
    assert dp_child_plan(1) == ("gloo", 2)
    assert dp_child_plan(2) == ("nccl", 2) and dp_child_plan(4) == ("nccl", 4) and dp_child_plan(8) == ("nccl", 4)
    assert dp_child_plan(16) == ("nccl", 4)


def test_gpu_local_cpus_reads_the_pci_devices_numa_cores(tmp_path):
    """bench.gpu_local_cpus: AMD display / accelerator devices of a (fake) sysfs tree in PCI bus order -> the core list of the i-th one;
    other vendors and other device classes are skipped; a missing index gives None (no pinning)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    pci = tmp_path / "pci"
    drm = tmp_path / "drm"
    drm.mkdir()
    spec = [("card0", "0000:05:00.0", "0x1a03", "0x030000", "0-7"),            # the board's VGA chip
            ("card2", "0000:c3:00.0", "0x1002", "0x120000", "48-63,112-127"),
            ("card1", "0000:43:00.0", "0x1002", "0x038000", "0-15,64-79"),
            ("card3", "0000:e3:00.0", "0x1002", "0x040300", "0-3")]            # an audio function
    for card, addr, vendor, cls, cpus in spec:
        d = pci / addr
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n"); (d / "class").write_text(cls + "\n"); (d / "local_cpulist").write_text(cpus + "\n")
        (drm / card).mkdir()
        os.symlink(str(d), str(drm / card / "device"))
    assert bench.gpu_local_cpus(0, str(drm)) == list(range(0, 16)) + list(range(64, 80))
    assert bench.gpu_local_cpus(1, str(drm)) == list(range(48, 64)) + list(range(112, 128))
    assert bench.gpu_local_cpus(2, str(drm)) is None
    assert bench.pin_rank_to_gpu_numa_node(0, 1) is None                       # a single-GPU run is never pinned
