"""micro-benchmark of cp_upsample2x_bilinear_ac at the decoder's shapes, beside a plain device copy of the same bytes"""
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, W, Cc, ocs, ooff) in ((32, 32, 256, 256, 0), (32, 32, 256, 320, 0), (32, 32, 64, 320, 256), (16, 16, 256, 384, 0), (16, 16, 128, 384, 256)):
    x = torch.randn(B, H, W, Cc, device=dev).to(torch.bfloat16)
    out = torch.empty(B, 2 * H, 2 * W, ocs, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _abi.check(lib.cp_upsample2x_bilinear_ac(st, 1, x.data_ptr(), out.data_ptr(), B, H, W, Cc, Cc, 0, ocs, ooff))
    us = timeit(run)
    by = B * H * W * Cc * 2 * 5
    src = torch.empty(by // 2 // 5 * 4, device=dev, dtype=torch.bfloat16)
    dst = torch.empty_like(src)
    usc = timeit(lambda: dst.copy_(src))
    print("%2dx%2d C=%3d -> cs=%3d+%3d B=%d: %7.1f us  %5.2f TB/s   (copy of the written bytes: %7.1f us %5.2f TB/s)"
          % (H, W, Cc, ocs, ooff, B, us, by / us / 1e6, usc, by / 5 * 8 / usc / 1e6), flush=True)
