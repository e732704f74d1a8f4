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
