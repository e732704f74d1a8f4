#!/usr/bin/env python3
"""Micro-benchmark of cp_basicblock_fused (18-channel HRNet branch at 64x64) -- A/B tool for kernel work.
   CP_BENCH_LIB=<alt .so> python tools/bb_bench.py [--batch 256] [--channels 18] [--size 64]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from checkerpose_amd import _abi  # noqa: E402
from checkerpose_amd._abi import ACT_RELU, CP_BF16, CpConvDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--channels", type=int, default=18)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    if os.environ.get("CP_BENCH_LIB"):
        _abi.LIB_PATH = os.environ["CP_BENCH_LIB"]
    lib = _abi.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B, H, Cc = a.batch, a.size, a.channels
    cp = (Cc + 7) // 8 * 8
    x = (torch.randn(B, H, H, cp, device=dev) * 0.5).to(torch.bfloat16)
    x[..., Cc:] = 0
    out = torch.empty_like(x)
    w = [(torch.randn(Cc, Cc, 3, 3, device=dev) * 0.1).contiguous() for _ in range(2)]
    nb = lib.cp_packed_halo_weight_bytes(CP_BF16, Cc, cp)
    pw = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
    _abi.check(lib.cp_pack_conv3x3_rows_weight(st, CP_BF16, w[0].data_ptr(), Cc, Cc, cp, pw[0].data_ptr()))
    _abi.check(lib.cp_pack_conv3x3_halo_weight(st, CP_BF16, w[1].data_ptr(), Cc, Cc, cp, pw[1].data_ptr()))
    aff = [torch.ones(32, device=dev) if i % 2 == 0 else torch.zeros(32, device=dev) for i in range(4)]
    d = CpConvDesc()
    d.dtype, d.out_f32, d.B, d.H, d.W = CP_BF16, 0, B, H, H
    d.Cin, d.in_cstride, d.in_coff = cp, cp, 0
    d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = 3, 3, 1, 1, H, H, cp, ACT_RELU, 0.0
    d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, H * H * cp, H * cp, cp, 1

    def run():
        _abi.check(lib.cp_basicblock_fused(st, C.byref(d), x.data_ptr(), pw[0].data_ptr(), aff[0].data_ptr(), aff[1].data_ptr(),
                                           pw[1].data_ptr(), aff[2].data_ptr(), aff[3].data_ptr(), out.data_ptr()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    nbytes = 2 * B * H * H * cp * 2
    print("basicblock_fused B=%d %dx%d C=%d: %.1f us  %.0f GB/s (alg)  %.1f TFLOP/s (unpadded)" %
          (B, H, H, Cc, us, nbytes / us / 1e3, 2 * 2 * B * H * H * 9 * Cc * Cc / us / 1e6))


if __name__ == "__main__":
    main()
