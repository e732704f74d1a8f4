"""In-kernel phase clock of mlp_pair_fused (workgroup 0): needs a -DCP_DEBUG_KNOBS build of mlp_fused.hip, e.g.
CHECKERPOSE_AMD_LIB=build/lib_knobs.so python tools/mlp_stamps.py [B]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CP_BF16
lib = _abi.load()
raw = C.CDLL(lib._name)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = ["L1 DMA issue", "L1 MFMA loop", "L1 epilogue", "L1 DMA wait", "barrier", "L2 compute", "L2 lgkm wait", "-"]
for N, Cin in ((4096, 512), (512, 512)):
    x = torch.randn(B, N, Cin, device=dev).to(torch.bfloat16)
    pk = []
    for ci in (Cin, 256):
        w = (torch.randn(256, ci, device=dev) * 0.05).contiguous()
        buf = torch.empty(lib.cp_packed_gemm_weight_bytes(CP_BF16, 256, ci), dtype=torch.uint8, device=dev)
        _abi.check(lib.cp_pack_gemm_weight(st, CP_BF16, w.data_ptr(), 256, ci, ci, buf.data_ptr()))
        pk.append(buf)
    b1, b2 = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    out = torch.empty(B, N, 256, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _abi.check(lib.cp_mlp_pair_fused(st, x.data_ptr(), Cin, 0, Cin, B, N, pk[0].data_ptr(), b1.data_ptr(), 0.01,
                                         pk[1].data_ptr(), b2.data_ptr(), 0.01, out.data_ptr(), 256, 0))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 96)()
    fn = raw.cp_debug_mlp_pair_stamps
    fn.restype = C.c_int
    assert fn(buf) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(12, 8)
    iters = (B * N // 32 + 255) // 256
    print("N=%d Cin=%d: %d iterations of workgroup 0; s_memtime ticks (100 MHz) per wave, sums over the launch" % (N, Cin, iters))
    for grp, rows, ks in (("layer-1 waves 0-7", slice(0, 8), (0, 1, 2, 3, 4)), ("layer-2 waves 8-11", slice(8, 12), (5, 6, 4))):
        tot = a[rows][:, list(ks)].sum(1).mean()
        print("  %s: total %.0f ticks = %.1f us (%.2f us per iteration)" % (grp, tot, tot / 100.0, tot / 100.0 / iters))
        for k in ks:
            print("    %-14s mean %8.0f (%.1f %%)  min %8.0f max %8.0f" % (names[k], a[rows, k].mean(), 100 * a[rows, k].mean() / tot, a[rows, k].min(), a[rows, k].max()))
