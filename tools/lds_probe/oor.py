"""LDS accesses beyond the workgroup's allocation on gfx950 (see oor.hip).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC oor.hip -o oor.so"""
import ctypes as C, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "oor.so"))
lib.oor_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for nbytes in (160 * 1024, 157 * 1024 + 512, 64 * 1024, 8 * 1024):
    out = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = lib.oor_run(st, out.data_ptr(), nbytes)
    torch.cuda.synchronize()
    o = out.tolist()
    print("LDS %6d B: rc %d | in-range dwords changed by the out-of-range writes: %d | out-of-range b32 reads != 0: %d of 512 | b128: %d of 512" % (nbytes, rc, o[0], o[1], o[2]))
