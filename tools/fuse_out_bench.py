"""micro-benchmark of cp_hr_fuse_out at the four HRNet-W18 stage-4 source branches"""
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CpFuseConv
lib = _abi.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CASES = [(18, 64, [(36, 3), (18, 3), (18, 3)]), (36, 32, [(18, 1), (72, 3), (36, 3)]), (72, 16, [(18, 1), (36, 1), (144, 3)]),
         (144, 8, [(18, 1), (36, 1), (72, 1)]),
         (18, 64, [(8, 1)]), (18, 64, [(36, 3)]), (18, 64, [(18, 3)]), (36, 32, [(72, 3)]), (72, 16, [(144, 3)]), (72, 16, [(8, 1)]), (144, 8, [(8, 1)])]
for Cc, H, convs in CASES:
    cp = (Cc + 7) // 8 * 8
    x = torch.randn(B, H, H, cp, device=dev).to(torch.bfloat16)
    arr = (CpFuseConv * len(convs))()
    keep, fl = [], 0
    st = torch.cuda.current_stream().cuda_stream
    for i, (Co, k) in enumerate(convs):
        kind = 1 if k == 3 else 0
        ocp = (Co + 7) // 8 * 8
        pw = torch.empty(lib.cp_hr_fuse_out_weight_bytes(cp, ocp, kind), dtype=torch.uint8, device=dev)
        w = (torch.randn(Co, Cc, k, k, device=dev) * 0.05).contiguous()
        _abi.check(lib.cp_pack_hr_fuse_out_weight(st, w.data_ptr(), Co, Cc, cp, ocp, kind, pw.data_ptr()))
        aff = torch.ones(2, lib.cp_hr_fuse_out_affine_floats(ocp), device=dev)
        out = torch.empty(B, H >> kind, H >> kind, ocp, device=dev, dtype=torch.bfloat16)
        arr[i].packed_w, arr[i].affine, arr[i].out = pw.data_ptr(), aff.data_ptr(), out.data_ptr()
        arr[i].kind, arr[i].Cout, arr[i].out_cphys, arr[i].relu = kind, Co, ocp, 0
        keep += [pw, w, aff, out]
        fl += 2 * B * (H >> kind) ** 2 * k * k * Cc * Co
    run = lambda: _abi.check(lib.cp_hr_fuse_out(st, x.data_ptr(), B, H, H, cp, len(convs), arr))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("C=%3d %2dx%2d B=%d %s: %7.1f us  %6.1f TF/s" % (Cc, H, H, B, convs, us, fl / us / 1e6), flush=True)
