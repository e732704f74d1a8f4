"""Data-parallel sharding helpers: one process per GPU, crops are independent units (eval-mode BatchNorm has no
cross-sample term), so the forward needs NO collective -- ranks only agree on who owns which crops and on the
step time.  The same functions run under `gloo` on CPU in the tests and `nccl` (= RCCL over xGMI)
on the GPU node."""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world):
    """Contiguous batch shard [lo, hi) of this rank; the first `global_batch % world` ranks take one extra crop."""
    if not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device=None, group=None):
    """Step time of the slowest rank (the whole-job time of a data-parallel step)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_over_ranks(value, device=None, group=None):
    """[rank 0's value, rank 1's value, ...] on every rank (a bench line checks itself with it: `ranks_seen`, `per_rank_ms`)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]


def aggregate_crops_per_sec(local_crops, seconds, device=None, group=None):
    """Whole-job throughput: all crops processed by all ranks / slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_crops / seconds
    n = torch.tensor([float(local_crops)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return float(n.item()) / max_over_ranks(seconds, device, group)


def gather_outputs(local, device=None, group=None):
    """Optional: concatenate per-rank output shards (B_r, ...) in rank order on every rank (ragged B_r allowed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)], 0)


def allreduce_gradients_(tensors, bucket_bytes=64 << 20, average=True, group=None):
    """The ONE collective of the data-parallel training step: sum (or mean) the gradient tensors of
    all ranks, in place.  Gradients are packed into flat buckets of ~`bucket_bytes` and each bucket is one
    all-reduce: on xGMI (point-to-point links, ring collectives are per-link bound) a few large messages beat
    hundreds of per-parameter ones; ~21 M parameters = 85 MB fp32 -> two 64 MiB buckets.  Launch order is
    deterministic (list order), so every rank issues the same sequence.  Works under `gloo` (CPU tests) and `nccl`
    (= RCCL on ROCm).  The backward that produces the gradients is row N1 (next round); this is its comm half."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    buckets, cur, cur_bytes = [], [], 0
    for t in tensors:
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or t.dtype != cur[0].dtype or t.device != cur[0].device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    for b in buckets:
        if len(b) == 1 and b[0].is_contiguous():          # a single flat tensor: reduce it in place, no staging copy
            dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=group)
            if average:
                b[0].div_(world)
            continue
        flat = torch.cat([t.reshape(-1) for t in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for t in b:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return len(buckets)


def plan_gradient_buckets(slots, total, done_call, first_call, n_calls, nbuckets=4):
    """slots: parameter name -> element offset into the flat gradient buffer of `total` elements (module order);
    done_call[name] = index of the first backward launch by which that parameter's gradient is final (absent: never written).
    Returns [(call_lo, call_hi, elem_lo, elem_hi)]: ~equal contiguous buckets, cut from the buffer's tail (the backward runs
    head first, so the tail -- decoder / refinement parameters -- is final long before the backbone's gradients), with the
    launches [first_call, n_calls) split into as many consecutive segments such that bucket k is final after segment k."""
    keys = sorted(slots, key=slots.get)
    starts = [slots[k] for k in keys]
    ends = starts[1:] + [total]
    target = max(total // max(nbuckets, 1), 1)
    buckets, hi, done, size = [], total, first_call, 0
    for k, lo_k, hi_k in reversed(list(zip(keys, starts, ends))):
        done = max(done, done_call.get(k, first_call))
        size += hi_k - lo_k
        if size >= target and len(buckets) < nbuckets - 1 and lo_k > 0:
            buckets.append((done, lo_k, hi))
            hi, done, size = lo_k, first_call, 0
    buckets.append((done, 0, hi))
    buckets.sort()
    segs, lo = [], first_call
    for i, (d, a, b) in enumerate(buckets):
        d = n_calls if i == len(buckets) - 1 else min(max(d, lo), n_calls)
        segs.append((lo, d, a, b))
        lo = d
    return segs


def backward_with_bucketed_allreduce_(flat, segments, run_segment, average=True, group=None):
    """The data-parallel backward: `segments` = [(lo, hi, a, b)] in launch order; run_segment(k, lo, hi) enqueues the backward
    launches [lo, hi) on the current stream, after which flat[a:b] is final.  Each bucket's all-reduce is issued
    asynchronously right behind its segment, so it travels (RCCL ring over xGMI / gloo in the tests) while the later
    segments compute; all are awaited at the end and the buffer is averaged in place -- no staging copy of the 85 MB
    buffer.  Every rank issues the same collectives in the same order.  Returns the number of collectives."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    works = []
    for k, (lo, hi, a, b) in enumerate(segments):
        run_segment(k, lo, hi)
        if on and b > a:
            works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if on and average:
        flat.div_(dist.get_world_size(group))
    return len(works)
