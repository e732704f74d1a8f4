"""micro-benchmark of cp_gemm_rows (row GEMM / Linear) at the refinement MLP shapes"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpConvDesc
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M = 131072
for (K, N) in ((256, 256), (512, 256), (320, 256), (256, 512), (64, 512)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, 1, 1, device=dev) * 0.05).contiguous()
    pw = torch.empty(lib.cp_packed_gemm_weight_bytes(1, N, K), dtype=torch.uint8, device=dev)
    _abi.check(lib.cp_pack_gemm_weight(st, 1, w.data_ptr(), N, K, K, pw.data_ptr()))
    sc = torch.ones(N, device=dev); sh = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = 1, 0, M // 512, 1, 512
    d.Cin, d.in_cstride, d.in_coff = K, K, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 1, 1, 1, 0, 1, 512, N, 2, 0.01
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, 512 * N, 512 * N, N, 1
    run = lambda: _abi.check(lib.cp_gemm_rows(st, C.byref(d), x.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2 * M * K * N
    by = M * (K + N) * 2
    print("M=%d K=%3d N=%3d: %7.1f us  %6.1f TF/s  %5.2f TB/s" % (M, K, N, us, fl / us / 1e6, by / us / 1e6), flush=True)
