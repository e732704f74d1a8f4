"""micro-benchmark of cp_hr_branch_chain (kernel work; KNOBS build honours CP_CHAIN_DBG)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for (Cc, H, W) in ((18, 64, 64), (36, 32, 32), (72, 16, 16), (144, 8, 8)):
    cp = (Cc + 7) // 8 * 8
    for nsrc in (1, 3):
        srcs = [torch.randn(B, H, W, cp, device=dev).to(torch.bfloat16) for _ in range(nsrc)]
        blob = torch.zeros(lib.cp_hr_chain_weight_bytes(Cc, H, W), dtype=torch.uint8, device=dev)
        w = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05).contiguous()
        for i in range(8):
            if lib.cp_version() >= 201:
                _abi.check(lib.cp_pack_hr_chain_weight(torch.cuda.current_stream().cuda_stream, w.data_ptr(), None, Cc, H, W, i, blob.data_ptr()))
            else:       # A/B against an older build (CHECKERPOSE_AMD_LIB): the pack call had no scale argument
                import ctypes
                lib.cp_pack_hr_chain_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
                _abi.check(lib.cp_pack_hr_chain_weight(torch.cuda.current_stream().cuda_stream, w.data_ptr(), Cc, H, W, i, blob.data_ptr()))
        n = lib.cp_hr_chain_affine_floats(Cc, H, W)
        aff = torch.zeros(8, 2, n, device=dev); aff[:, 0, :Cc] = 0.5
        out = torch.empty(B, H, W, cp, device=dev, dtype=torch.bfloat16)
        arr_p = (C.c_void_p * 4)(*([s.data_ptr() for s in srcs] + [None] * (4 - nsrc)))
        arr_s = (C.c_int32 * 4)(0, 0, 0, 0)
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: _abi.check(lib.cp_hr_branch_chain(st, B, Cc, H, W, nsrc, arr_p, arr_s, 1, blob.data_ptr(), aff.data_ptr(), out.data_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 8 * 2 * B * H * W * 9 * Cc * Cc
        print("C=%3d %2dx%2d nsrc=%d B=%d: %7.1f us  %6.1f TF/s" % (Cc, H, W, nsrc, B, us, fl / us / 1e6), flush=True)
