"""Differentiable fused graph ops: torch.autograd.Function wrappers whose forward AND
backward are the library's HIP kernels.  Channels-last operands as everywhere in the engine.

  edgeconv_aggregate(pq, idx, slope)            -- cp_edgeconv_gather_max / cp_edgeconv_gather_max_bwd
  index2feat_gather(patches, x_id, y_id, mask)  -- cp_index2feat_gather  / cp_index2feat_gather_bwd
"""
import torch

from . import _abi
from ._abi import CP_BF16, CP_F32


def _dt(t):
    if t.dtype == torch.float32:
        return CP_F32
    if t.dtype == torch.bfloat16:
        return CP_BF16
    raise TypeError("fp32 or bf16 tensors only")


def reverse_graph(idx):
    """idx (G,N,K) or (N,K) int -> (rev_ptr (G,N+1) int32, rev_edge (G,N*K) int32): for every node j the flat edge
    ids i*K+k with idx[i,k] == j, in increasing edge order (stable sort).  Static per object model -- build once."""
    idx3 = idx if idx.dim() == 3 else idx[None]
    G, N, K = idx3.shape
    flat = idx3.reshape(G, N * K).long()
    rev_edge = torch.argsort(flat, dim=1, stable=True).int()
    counts = torch.zeros(G, N, dtype=torch.long, device=idx.device)
    counts.scatter_add_(1, flat, torch.ones_like(flat))
    rev_ptr = torch.zeros(G, N + 1, dtype=torch.int32, device=idx.device)
    rev_ptr[:, 1:] = counts.cumsum(1).int()
    return rev_ptr.contiguous(), rev_edge.contiguous()


class _EdgeConvAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pq, idx, rev_ptr, rev_edge, graph_ids, slope):
        if not pq.is_cuda:
            raise RuntimeError("checkerpose_amd.train_ops: CUDA/HIP tensors required (no CPU fallback)")
        lib = _abi.load()
        pq = pq.contiguous()
        B, N, C2 = pq.shape
        C = C2 // 2
        idx3 = idx if idx.dim() == 3 else idx[None]
        G, _, K = idx3.shape
        out = torch.empty(B, N, C, dtype=pq.dtype, device=pq.device)
        st = torch.cuda.current_stream(pq.device).cuda_stream
        _abi.check(lib.cp_edgeconv_gather_max(st, _dt(pq), pq.data_ptr(), idx3.data_ptr(),
                                              graph_ids.data_ptr() if graph_ids is not None else None, out.data_ptr(),
                                              B, N, K, C, G, C, 0, float(slope)), "cp_edgeconv_gather_max")
        ctx.save_for_backward(pq, idx3, rev_ptr, rev_edge)
        ctx.graph_ids, ctx.slope = graph_ids, float(slope)
        return out

    @staticmethod
    def backward(ctx, gout):
        pq, idx3, rev_ptr, rev_edge = ctx.saved_tensors
        lib = _abi.load()
        B, N, C2 = pq.shape
        C = C2 // 2
        G, _, K = idx3.shape
        g = gout.float().contiguous()
        dpq = torch.empty(B, N, C2, dtype=torch.float32, device=pq.device)
        ws = torch.empty(lib.cp_edgeconv_bwd_workspace_bytes(B, N, C), dtype=torch.uint8, device=pq.device)
        st = torch.cuda.current_stream(pq.device).cuda_stream
        gi = ctx.graph_ids
        _abi.check(lib.cp_edgeconv_gather_max_bwd(st, _dt(pq), pq.data_ptr(), idx3.data_ptr(), rev_ptr.data_ptr(),
                                                  rev_edge.data_ptr(), gi.data_ptr() if gi is not None else None,
                                                  g.data_ptr(), dpq.data_ptr(), ws.data_ptr(), B, N, K, C, G, C, 0,
                                                  ctx.slope), "cp_edgeconv_gather_max_bwd")
        return dpq.to(pq.dtype), None, None, None, None, None


def edgeconv_aggregate(pq, idx, slope=0.2, rev=None, graph_ids=None):
    """out[b,i,c] = leaky(max_k P'[b, idx[i,k], c] + Q'[b,i,c]); pq (B,N,2C) = [P'|Q'], idx (N,K)/(G,N,K) int32."""
    idx = idx.int().contiguous()
    rev_ptr, rev_edge = rev if rev is not None else reverse_graph(idx)
    return _EdgeConvAggregate.apply(pq, idx, rev_ptr, rev_edge, graph_ids, slope)


class _Index2Feat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, x_id, y_id, mask, k):
        if not patches.is_cuda:
            raise RuntimeError("checkerpose_amd.train_ops: CUDA/HIP tensors required (no CPU fallback)")
        lib = _abi.load()
        patches = patches.contiguous()
        B, Hp, Wp, E = patches.shape
        N = x_id.shape[1]
        x32, y32 = x_id.int().contiguous(), y_id.int().contiguous()
        m = mask.float().contiguous()
        out = torch.empty(B, N, 4 * E, dtype=patches.dtype, device=patches.device)
        st = torch.cuda.current_stream(patches.device).cuda_stream
        _abi.check(lib.cp_index2feat_gather(st, _dt(patches), patches.data_ptr(), x32.data_ptr(), y32.data_ptr(), m.data_ptr(),
                                            out.data_ptr(), B, N, Hp, Wp, E, k, 4 * E, 0), "cp_index2feat_gather")
        ctx.save_for_backward(x32, y32, m)
        ctx.shape, ctx.k, ctx.dtype = (B, Hp, Wp, E), k, patches.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        x32, y32, m = ctx.saved_tensors
        lib = _abi.load()
        B, Hp, Wp, E = ctx.shape
        N = x32.shape[1]
        g = gout.float().contiguous()
        dp = torch.empty(B, Hp, Wp, E, dtype=torch.float32, device=g.device)
        st = torch.cuda.current_stream(g.device).cuda_stream
        _abi.check(lib.cp_index2feat_gather_bwd(st, g.data_ptr(), x32.data_ptr(), y32.data_ptr(), m.data_ptr(), dp.data_ptr(),
                                                B, N, Hp, Wp, E, ctx.k, 4 * E, 0), "cp_index2feat_gather_bwd")
        return dp.to(ctx.dtype), None, None, None, None


def index2feat_gather(patches, x_id, y_id, mask, k=2):
    """patches (B,Hp,Wp,E) channels-last; ids (B,N); mask (B,N) -> (B,N,4E), taps sf1..sf4 (pipeline.py:158-162)."""
    return _Index2Feat.apply(patches, x_id, y_id, mask, k)
