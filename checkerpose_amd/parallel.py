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
