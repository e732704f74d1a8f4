#!/usr/bin/env python3
"""HBM calibration on this box: what a plain streaming copy / add reaches (context for the roofline fractions)."""
import torch
dev = torch.device("cuda:0")
for mb in (64, 256, 1024):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev).to(torch.bfloat16); b = torch.randn(n, device=dev).to(torch.bfloat16); c = torch.empty_like(a)
    for name, fn, nbytes in (("copy", lambda: c.copy_(a), 2 * n * 2), ("add", lambda: torch.add(a, b, out=c), 3 * n * 2)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-5s %5d MB tensors: %8.1f us  %6.0f GB/s" % (name, mb, us, nbytes / us / 1e3))
