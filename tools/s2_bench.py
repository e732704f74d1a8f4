"""micro-benchmark of cp_conv3x3_s2_small at HRNet transition1[1] (256 -> 36, 64 x 64 -> 32 x 32)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpConvDesc
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, Cin, H, W, Cout = 256, 256, 64, 64, 36
ocp = 40
x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02).contiguous()
pw = torch.empty(lib.cp_conv3x3_s2_small_weight_bytes(Cin, ocp), dtype=torch.uint8, device=dev)
_abi.check(lib.cp_pack_conv3x3_s2_small_weight(st, w.data_ptr(), Cout, Cin, Cin, ocp, pw.data_ptr()))
sc = torch.ones(48, device=dev); sh = torch.zeros(48, device=dev)
out = torch.empty(B, H // 2, W // 2, ocp, device=dev, dtype=torch.bfloat16)
d = CpConvDesc()
d.dtype, d.out_f32, d.B, d.H, d.W = 1, 0, B, H, W
d.Cin, d.in_cstride, d.in_coff = Cin, Cin, 0
d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 3, 3, 2, 1, H // 2, W // 2, ocp, 1, 0.0
d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, (H // 2) * (W // 2) * ocp, (W // 2) * ocp, ocp, 1
run = lambda: _abi.check(lib.cp_conv3x3_s2_small(st, C.byref(d), x.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr()))
for rep in range(3):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("s2 conv 256->36 64x64 B=%d: %7.1f us  %5.2f TB/s read" % (B, us, B * H * W * Cin * 2 / us / 1e6), flush=True)
