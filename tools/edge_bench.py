"""micro-benchmark of cp_edgeconv_fused_t (one EdgeConv layer, N = 512, K = 20, 256 -> 256) in bf16 and in half, interleaved"""
import os
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, K = 512, 20
for Cin, Cout, dt in ((256, 256, 1), (256, 256, 2), (256, 256, 1), (256, 256, 2), (64, 64, 1), (64, 64, 2)):       # dt: CP_BF16 = 1, CP_F16 = 2
    x = torch.randn(B, N, Cin, device=dev).to(torch.bfloat16 if dt == 1 else torch.float16)
    wpq = (torch.randn(2 * Cout, Cin, device=dev) * 0.05).contiguous()
    pw = torch.empty(lib.cp_edgeconv_fused_weight_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    _abi.check(lib.cp_pack_edgeconv_fused_weight_t(st, dt, wpq.data_ptr(), Cin, Cout, pw.data_ptr()))
    sc = torch.ones(2 * Cout, device=dev); sh = torch.zeros(2 * Cout, device=dev)
    idx = torch.randint(0, N, (1, N, K), device=dev, dtype=torch.int32)
    if os.environ.get("EDGE_BENCH_SCHED", "1") != "0":      # the order the engine hands in (graph_sched.py); 0: as drawn
        from checkerpose_amd.graph_sched import schedule_neighbours
        o, before, after = schedule_neighbours(idx.cpu().numpy())
        print("scheduled neighbour lists: residue clashes %d -> %d of %d reads" % (before, after, o.size))
        idx = torch.from_numpy(o).to(dev)
    out = torch.empty(B, N, Cout, device=dev, dtype=torch.bfloat16)
    run = lambda: _abi.check(lib.cp_edgeconv_fused_t(st, dt, x.data_ptr(), Cin, 0, pw.data_ptr(), sc.data_ptr(), sh.data_ptr(), idx.data_ptr(), None,
                                                  out.data_ptr(), Cout, 0, B, N, K, Cin, Cout, 1, 0.2))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%s Cin=%3d Cout=%3d B=%d: %7.1f us  %6.1f TF/s" % ("bf16" if dt == 1 else "half", Cin, Cout, B, us, 2 * B * N * Cin * 2 * Cout / us / 1e6), flush=True)
