"""micro-benchmark of cp_mlp_pair_fused (pre_graph_module: Cin -> 256 -> 256 over B x N rows): us per launch, TFLOP/s, GB/s"""
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd._abi import CP_BF16
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for N, Cin in ((4096, 512), (512, 512), (4096, 320)):
    x = torch.randn(B, N, Cin, device=dev).to(torch.bfloat16)
    pk = []
    for ci in (Cin, 256):
        w = (torch.randn(256, ci, device=dev) * 0.05).contiguous()
        buf = torch.empty(lib.cp_packed_gemm_weight_bytes(CP_BF16, 256, ci), dtype=torch.uint8, device=dev)
        _abi.check(lib.cp_pack_gemm_weight(st, CP_BF16, w.data_ptr(), 256, ci, ci, buf.data_ptr()))
        pk.append(buf)
    b1, b2 = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    out = torch.empty(B, N, 256, device=dev, dtype=torch.bfloat16)
    run = lambda: _abi.check(lib.cp_mlp_pair_fused(st, x.data_ptr(), Cin, 0, Cin, B, N, pk[0].data_ptr(), b1.data_ptr(), 0.01,
                                                   pk[1].data_ptr(), b2.data_ptr(), 0.01, out.data_ptr(), 256, 0))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * B * N * (Cin * 256 + 256 * 256)
    by = B * N * (Cin + 256) * 2.0
    print("B=%d N=%d Cin=%d: %8.1f us  %7.1f TF/s  %6.0f GB/s  checksum %.4f" % (B, N, Cin, us, fl / us / 1e6, by / us / 1e3, float(out.float().abs().mean())))

# ---- cp_mlp_query_fused (MLP_QueryNet 256 -> 256 -> 64 -> 2 over B x N rows)
for N in (4096, 512):
    x = torch.randn(B, N, 256, device=dev).to(torch.bfloat16)
    pk = []
    for co, ci in ((256, 256), (64, 256)):
        w = (torch.randn(co, ci, device=dev) * 0.05).contiguous()
        buf = torch.empty(lib.cp_packed_gemm_weight_bytes(CP_BF16, co, ci), dtype=torch.uint8, device=dev)
        _abi.check(lib.cp_pack_gemm_weight(st, CP_BF16, w.data_ptr(), co, ci, ci, buf.data_ptr()))
        pk.append(buf)
    ones = [torch.ones(256, device=dev), torch.ones(64, device=dev)]
    zer = [torch.zeros(256, device=dev), torch.zeros(64, device=dev)]
    w3 = (torch.randn(2, 64, device=dev) * 0.1).contiguous()
    b3 = torch.zeros(2, device=dev)
    out = torch.zeros(B, 2, N, device=dev)
    run = lambda: _abi.check(lib.cp_mlp_query_fused(st, x.data_ptr(), 256, 0, B, N, pk[0].data_ptr(), ones[0].data_ptr(), zer[0].data_ptr(), 0.01,
                                                    pk[1].data_ptr(), ones[1].data_ptr(), zer[1].data_ptr(), 0.01, w3.data_ptr(), b3.data_ptr(),
                                                    out.data_ptr(), 0, 2 * N, 1, N))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("query B=%d N=%d: %8.1f us  %6.0f GB/s  checksum %.4f" % (B, N, us, B * N * 256 * 2.0 / us / 1e3, float(out.abs().mean())))
