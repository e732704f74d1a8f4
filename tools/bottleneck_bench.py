"""micro-benchmark of cp_bottleneck_fused (HRNet layer1, 64 x 64 x 256, identity and projection shortcut)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpConvDesc
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = 64


def pack(co, ci, r):
    w = (torch.randn(co, ci, r, r, device=dev) * 0.05).contiguous()
    pw = torch.empty(lib.cp_packed_weight_bytes(1, co, ci, r, r), dtype=torch.uint8, device=dev)
    _abi.check(lib.cp_pack_conv_weight(st, 1, w.data_ptr(), co, ci, r, r, ci, 0, 0, None, co, pw.data_ptr()))
    return pw


for cin in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (256, 64)):
    ds = cin == 64
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    out = torch.empty(B, H, H, 256, device=dev, dtype=torch.bfloat16)
    ws = [pack(64, cin, 1), pack(64, 64, 3), pack(256, 64, 1)] + ([pack(256, cin, 1)] if ds else [])
    aff = [torch.ones(n, device=dev) for n in (64, 64, 64, 64, 256, 256, 256, 256)]
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = 1, 0, B, H, H
    d.Cin, d.in_cstride, d.in_coff = cin, cin, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 1, 1, 1, 0, H, H, 256, 1, 0.0
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, H * H * 256, H * 256, 256, 1
    args = (ws[0].data_ptr(), aff[0].data_ptr(), aff[1].data_ptr(), ws[1].data_ptr(), aff[2].data_ptr(), aff[3].data_ptr(),
            ws[2].data_ptr(), aff[4].data_ptr(), aff[5].data_ptr(),
            ws[3].data_ptr() if ds else None, aff[6].data_ptr() if ds else None, aff[7].data_ptr() if ds else None)
    run = lambda: _abi.check(lib.cp_bottleneck_fused(st, C.byref(d), x.data_ptr(), *args, out.data_ptr()))
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
        by = B * H * H * (cin + 256) * 2
        print("Cin=%3d B=%d: %7.1f us  %5.2f TB/s" % (cin, B, us, by / us / 1e6), flush=True)
