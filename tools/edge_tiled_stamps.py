"""In-kernel phase clock of edgeconv_tiled2 (workgroup 0): needs a `-DCP_DEBUG_KNOBS` build of edgeconv_tiled.hip, e.g.
CHECKERPOSE_AMD_LIB=build/lib_knobs.so python tools/edge_tiled_stamps.py [B]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd.graph_sched import tile_schedule
from checkerpose_amd.model.init import knn
from checkerpose_amd.synthetic import lm_p3d
lib = _abi.load()
raw = C.CDLL(lib._name) if hasattr(lib, "_name") else lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, K = 4096, 20
P = lm_p3d(N)[[0, 4, 13]]
sc = tile_schedule(knn(P, K).numpy(), P.numpy())
halo, nbr = torch.from_numpy(sc["halo"]).contiguous().to(dev), torch.from_numpy(sc["nbr"]).contiguous().to(dev)
gids = torch.arange(B, dtype=torch.int32, device=dev) % 3
names = ["prologue issue", "prologue wait", "DMA issue", "pass0 steps", "epilogue0", "pass1 to mid", "mid barrier", "pass1 rest", "epilogue1",
         "end barrier", "-", "-"]
for Cin, Cout in ((256, 256), (64, 64)):
    x = torch.randn(B, N, Cin, device=dev).to(torch.bfloat16)
    wpq = (torch.randn(2 * Cout, Cin, device=dev) * 0.05).contiguous()
    pf = torch.empty(lib.cp_edgeconv_fused_weight_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    pq = torch.empty(lib.cp_edgeconv_tiled_weight_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    _abi.check(lib.cp_pack_edgeconv_fused_weight(st, wpq.data_ptr(), Cin, Cout, pf.data_ptr()))
    _abi.check(lib.cp_pack_edgeconv_tiled_weight(st, wpq.data_ptr(), Cin, Cout, pq.data_ptr()))
    s_, t_ = torch.ones(2 * Cout, device=dev), torch.zeros(2 * Cout, device=dev)
    ktab = torch.empty(lib.cp_edgeconv_tiled_table_bytes(B, N, Cout), dtype=torch.uint8, device=dev)
    out = torch.empty(B, N, Cout, device=dev, dtype=torch.bfloat16)
    run = lambda: _abi.check(lib.cp_edgeconv_tiled(st, x.data_ptr(), Cin, 0, pf.data_ptr(), pq.data_ptr(), s_.data_ptr(), t_.data_ptr(),
                                                   halo.data_ptr(), nbr.data_ptr(), gids.data_ptr(), ktab.data_ptr(), out.data_ptr(), Cout, 0,
                                                   B, N, K, Cin, Cout, 3, int(sc["HPAD"]), 0.2))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 96)()
    fn = raw.cp_debug_edge_tiled_stamps
    fn.restype = C.c_int
    assert fn(buf) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(8, 12)
    tot = a.sum(1)
    print("Cin=%d: cycles per wave (s_memtime ticks), workgroup 0; total per wave: %s" % (Cin, np.round(tot).astype(int).tolist()))
    for k, nme in enumerate(names[:10]):
        print("  %-16s mean %8.0f  (%.1f %%)   min %8.0f max %8.0f" % (nme, a[:, k].mean(), 100 * a[:, k].mean() / tot.mean(), a[:, k].min(), a[:, k].max()))
