"""micro-benchmark of cp_edgeconv_tiled (one EdgeConv layer at N = 4096, K = 20, 256 -> 256, LM object graphs, patch schedule of
graph_sched.tile_schedule): the pair of launches, batch 32.  A/B of two builds: CHECKERPOSE_AMD_LIB=<other .so>."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd.graph_sched import tile_schedule
from checkerpose_amd.model.init import knn
from checkerpose_amd.synthetic import lm_p3d
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, K = 4096, 20
P = lm_p3d(N)[[0, 4, 13]]
sc = tile_schedule(knn(P, K).numpy(), P.numpy())
halo, nbr = torch.from_numpy(sc["halo"]).contiguous().to(dev), torch.from_numpy(sc["nbr"]).contiguous().to(dev)
gids = torch.arange(B, dtype=torch.int32, device=dev) % 3
print("HPAD %d, halo rows per patch: max %d mean %.0f" % (sc["HPAD"], sc["halo_rows"].max(), sc["halo_rows"].mean()))
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
    run = lambda: _abi.check(lib.cp_edgeconv_tiled(st, x.data_ptr(), Cin, 0, pf.data_ptr(), pq.data_ptr(), s_.data_ptr(), t_.data_ptr(), halo.data_ptr(),
                                                   nbr.data_ptr(), gids.data_ptr(), ktab.data_ptr(), out.data_ptr(), Cout, 0, B, N, K, Cin, Cout, 3,
                                                   int(sc["HPAD"]), 0.2))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("Cin=%3d Cout=%3d B=%d N=%d: %7.1f us per layer (both launches)  %6.1f TF/s" % (Cin, Cout, B, N, us, 2 * B * N * Cin * 2 * Cout / us / 1e6), flush=True)
