"""micro-benchmark of cp_conv3x3_halo_up2x (decoder up_net[1..2] conv1) beside the plain halo conv at the same output size"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpConvDesc
lib = _abi.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = torch.cuda.current_stream().cuda_stream
for (Hs, Cin) in ((32, 512), (16, 768)):
    H = 2 * Hs
    x = torch.randn(B, Hs, Hs, Cin, device=dev).to(torch.bfloat16)
    xu = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(256, Cin, 3, 3, device=dev) * 0.02).contiguous()
    pw = torch.empty(lib.cp_packed_halo_weight_bytes(1, 256, Cin), dtype=torch.uint8, device=dev)
    _abi.check(lib.cp_pack_conv3x3_halo_weight(st, 1, w.data_ptr(), 256, Cin, Cin, pw.data_ptr()))
    sc = torch.ones(256, device=dev); sh = torch.zeros(256, device=dev)
    out = torch.empty(B, H, H, 256, device=dev, dtype=torch.bfloat16)
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = 1, 0, B, H, H
    d.Cin, d.in_cstride, d.in_coff = Cin, Cin, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 3, 3, 1, 1, H, H, 256, 1, 0.0
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, H * H * 256, H * 256, 256, 1
    runs = {"up2x": lambda: _abi.check(lib.cp_conv3x3_halo_up2x(st, C.byref(d), x.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr())),
            "plain": lambda: _abi.check(lib.cp_conv3x3_halo(st, C.byref(d), xu.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr()))}
    fl = 2 * B * H * H * 9 * Cin * 256
    for rep in range(2):
        for name, run in runs.items():
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            print("%2d->%2d Cin=%d %-5s: %8.1f us  %7.1f TF/s" % (Hs, H, Cin, name, us, fl / us / 1e6), flush=True)
