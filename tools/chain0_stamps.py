"""In-kernel phase clock of the 64 x 64 chain (workgroup 0, no tail; the pipelined form, or the band form when the library's file name contains
"band" = a -DCP_C0_BAND build): needs a -DCP_DEBUG_KNOBS build of hr_chain0.hip, e.g.
CHECKERPOSE_AMD_LIB=build/lib_knobs.so python tools/chain0_stamps.py [B]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
raw = C.CDLL(lib._name)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Cc, H, W, cp = 18, 64, 64, 24
names = (["loop", "wait input", "reads + MFMAs", "epilogue", "wait slot", "write + post", "-", "-"] if "band" not in _abi.LIB_PATH else ["prologue", "band MFMA+reads", "band epilogue", "band barrier", "write-back", "conv barrier", "tail", "-"])
for nsrc in (1, 3):
    srcs = [torch.randn(B, H, W, cp, device=dev).to(torch.bfloat16) for _ in range(nsrc)]
    blob = torch.zeros(lib.cp_hr_chain_weight_bytes(Cc, H, W), dtype=torch.uint8, device=dev)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05).contiguous()
    for i in range(8):
        _abi.check(lib.cp_pack_hr_chain_weight(st, w.data_ptr(), None, Cc, H, W, i, blob.data_ptr()))
    n = lib.cp_hr_chain_affine_floats(Cc, H, W)
    aff = torch.zeros(8, 2, n, device=dev)
    out = torch.empty(B, H, W, cp, device=dev, dtype=torch.bfloat16)
    arr_p = (C.c_void_p * 4)(*([s.data_ptr() for s in srcs] + [None] * (4 - nsrc)))
    arr_s = (C.c_int32 * 4)(0, 0, 0, 0)
    for _ in range(3):
        _abi.check(lib.cp_hr_branch_chain(st, B, Cc, H, W, nsrc, arr_p, arr_s, 1, blob.data_ptr(), aff.data_ptr(), out.data_ptr()))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    fn = raw.cp_debug_chain0_stamps
    fn.restype = C.c_int
    assert fn(buf) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
    tot = a.sum(1)
    print("nsrc=%d: clock ticks per wave, workgroup 0: %s" % (nsrc, np.round(tot).astype(int).tolist()))
    for k, nme in enumerate(names[:7]):
        print("  %-16s mean %8.0f (%.1f %%)   min %8.0f max %8.0f   per wave %s" % (nme, a[:, k].mean(), 100 * a[:, k].mean() / tot.mean(), a[:, k].min(), a[:, k].max(),
                                                                               np.round(a[:, k] / 1000).astype(int).tolist()))
