"""Does the Infinity Cache (MALL) keep a producer's output for the consumer launch?  Two row GEMMs in a chain (X -> Y -> Z, 256 -> 256
-> 256, M = 1 M rows: config #5's refinement MLPs at 256 crops x 4096 keypoints) run (a) whole: Y = 512 MB round-trips HBM, and
(b) in row chunks with ONE chunk-sized Y buffer reused by every chunk (the write -> read distance is a chunk, not the tensor)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpConvDesc
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M, K, N = 256 * 4096, 256, 256
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
z = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
w = (torch.randn(N, K, 1, 1, device=dev) * 0.05).contiguous()
pw = torch.empty(lib.cp_packed_gemm_weight_bytes(1, N, K), dtype=torch.uint8, device=dev)
_abi.check(lib.cp_pack_gemm_weight(st, 1, w.data_ptr(), N, K, K, pw.data_ptr()))
sc = torch.ones(N, device=dev); sh = torch.zeros(N, device=dev)


def gemm(src, dst, rows):
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = 1, 0, rows // 512, 1, 512
    d.Cin, d.in_cstride, d.in_coff = K, K, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 1, 1, 1, 0, 1, 512, N, 2, 0.01
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, 512 * N, 512 * N, N, 1
    _abi.check(lib.cp_gemm_rows(st, C.byref(d), src, pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, dst))


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for nchunk in (1, 2, 4, 8, 16, 32):
    rows = M // nchunk
    y = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)

    def run():
        for c in range(nchunk):
            gemm(x.data_ptr() + c * rows * K * 2, y.data_ptr(), rows)
            gemm(y.data_ptr(), z.data_ptr() + c * rows * N * 2, rows)
    us = timed(run)
    print("chunks %2d (Y buffer %4d MB): %8.1f us per chain   (compulsory X in + Z out = %.0f MB -> %.2f TB/s)" %
          (nchunk, rows * N * 2 >> 20, us, 2 * M * N * 2 / 1e6, 2 * M * N * 2 / us / 1e6), flush=True)
