"""How much of a small weight-gradient launch is fixed cost?  The HRNet-branch wgrad launches of the B=32 training step take
~23 us each inside the step (profiles/, rocprofv3).  Here the SAME launch (deferred form: partial tiles only) runs 200 times
back to back on a warm GPU, alone and interleaved with an unrelated kernel that evicts the instruction cache, timed with events."""
import ctypes as C
import torch
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpWgradDesc, CpWgradReduceItem, CP_BF16

lib = _abi.load()
dev = torch.device("cuda:0")
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def case(B, Cc, H):
    x = torch.randn(B, H, H, (Cc + 7) // 8 * 8, device=dev).bfloat16()
    dy = torch.randn_like(x)
    dw = torch.zeros(Cc, Cc, 3, 3, device=dev)
    d = CpWgradDesc()
    d.dtype, d.B, d.H, d.W, d.Ho, d.Wo = CP_BF16, B, H, H, H, H
    d.Cout, d.dy_cstride, d.dy_coff, d.Cin, d.x_cstride, d.x_coff = Cc, x.shape[-1], 0, Cc, x.shape[-1], 0
    d.R, d.S, d.stride, d.pad = 3, 3, 1, 1
    d.dw_base, d.dw_sco, d.dw_sci, d.dw_sr, d.dw_ss = 0, Cc * 9, 9, 3, 1
    it = CpWgradReduceItem()
    st = torch.cuda.current_stream().cuda_stream

    def go():
        _abi.check(lib.cp_conv2d_wgrad_deferred(st, C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(it)), "wgrad")
    return go, (x, dy, dw, d, it)


other_a = torch.randn(4096, 4096, device=dev).bfloat16()
for B, Cc, H in [(32, 36, 32), (32, 72, 16), (32, 144, 8), (32, 18, 64)]:
    go, keep = case(B, Cc, H)
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        go()
    e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 200 * 1e3
    # interleaved with a GEMM (different code, evicts I-cache / L2 lines): time of the pair minus the GEMM alone
    e0.record()
    for _ in range(50):
        other_a @ other_a
    e1.record()
    torch.cuda.synchronize()
    g = e0.elapsed_time(e1) / 50 * 1e3
    e0.record()
    for _ in range(50):
        other_a @ other_a
        go()
    e1.record()
    torch.cuda.synchronize()
    mixed = e0.elapsed_time(e1) / 50 * 1e3 - g
    print("wgrad3x3 B=%d C=%d %dx%d: %.1f us back to back, %.1f us behind an unrelated GEMM (%.0f us)" % (B, Cc, H, H, alone, mixed, g), flush=True)
