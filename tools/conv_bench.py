#!/usr/bin/env python3
"""Micro-benchmark of cp_conv2d_igemm on the shape classes of the CheckerPose forward (A/B tool for kernel work).
   python tools/conv_bench.py [--batch 64] [--dtype bf16|fp32|both] [--reps 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from checkerpose_amd import _abi  # noqa: E402
from checkerpose_amd._abi import CP_BF16, CP_F32, CpConvDesc  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, k, stride, pad, residual
    ("dec 512->256 3x3 @64", 64, 512, 256, 3, 1, 1, False),
    ("dec 256->256 3x3 @64", 64, 256, 256, 3, 1, 1, False),
    ("dec 768->256 3x3 @32", 32, 768, 256, 3, 1, 1, False),
    ("hr 18->18 3x3 @64 +res", 64, 18, 18, 3, 1, 1, True),
    ("hr 36->36 3x3 @32 +res", 32, 36, 36, 3, 1, 1, True),
    ("hr 72->72 3x3 @16 +res", 16, 72, 72, 3, 1, 1, True),
    ("hr 144->144 3x3 @8 +res", 8, 144, 144, 3, 1, 1, True),
    ("l1 64->256 1x1 @64 +res", 64, 64, 256, 1, 1, 0, True),
    ("l1 256->64 1x1 @64", 64, 256, 64, 1, 1, 0, False),
    ("l1 64->64 3x3 @64", 64, 64, 64, 3, 1, 1, False),
    ("stem 3->64 3x3 s2 @256", 256, 3, 64, 3, 2, 1, False),
    ("edge gemm 256->512 (N=512)", None, 256, 512, 1, 1, 0, False),
    ("mlp 256->256 (N=512)", None, 256, 256, 1, 1, 0, False),
    ("mlp 512->256 (N=512)", None, 512, 256, 1, 1, 0, False),
    ("incre 128->512 1x1 @16 +res", 16, 128, 512, 1, 1, 0, True),
    ("patch 256->64 k2 p1 @64", 64, 256, 64, 2, 1, 1, False),
    ("trans 256->18 3x3 @64", 64, 256, 18, 3, 1, 1, False),
]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None, help="substring filter on the shape name")
    ap.add_argument("--halo", type=int, default=1)
    ap.add_argument("--gemm", type=int, default=1)
    ap.add_argument("--gemm-min-k", type=int, default=16)
    ap.add_argument("--gemm-min-cout", type=int, default=96)
    a = ap.parse_args()
    if os.environ.get("CP_BENCH_LIB"):
        _abi.LIB_PATH = os.environ["CP_BENCH_LIB"]
    lib = _abi.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    dts = {"bf16": [CP_BF16], "fp32": [CP_F32], "both": [CP_BF16, CP_F32]}[a.dtype]
    print("%-37s %8s %10s %10s %9s" % ("shape (B=%d)" % a.batch, "dtype", "us", "TFLOP/s", "GB/s(alg)"))
    for name, H, Cin, Cout, k, stride, pad, has_res in SHAPES:
        if a.only and a.only not in name:
            continue
        for dt in dts:
            E, es, tdt = (8, 2, torch.bfloat16) if dt == CP_BF16 else (4, 4, torch.float32)
            B = a.batch
            if H is None:
                Hh, Ww = 1, 512
            else:
                Hh = Ww = H
            cin_p, cout_p = rup(Cin, E), rup(Cout, E)
            Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (Ww + 2 * pad - k) // stride + 1
            x = torch.randn(B, Hh, Ww, cin_p, device=dev).to(tdt)
            w = (torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5).contiguous()
            pw = torch.empty(lib.cp_packed_weight_bytes(dt, Cout, cin_p, k, k), dtype=torch.uint8, device=dev)
            _abi.check(lib.cp_pack_conv_weight(st, dt, w.data_ptr(), Cout, Cin, k, k, cin_p, 0, 0, None, Cout, pw.data_ptr()))
            sc = torch.ones(rup(Cout, 16), device=dev); sh = torch.zeros(rup(Cout, 16), device=dev)
            out = torch.empty(B, Ho, Wo, cout_p, device=dev, dtype=tdt)
            res = torch.randn(B, Ho, Wo, cout_p, device=dev).to(tdt) if has_res else None
            d = CpConvDesc()
            d.dtype, d.out_f32, d.B, d.H, d.W = dt, 0, B, Hh, Ww
            d.Cin, d.in_cstride, d.in_coff = cin_p, cin_p, 0
            d.R, d.S, d.stride, d.pad, d.Ho, d.Wo, d.Cout, d.act, d.slope = k, k, stride, pad, Ho, Wo, cout_p, 1, 0.0
            d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = 0, Ho * Wo * cout_p, Wo * cout_p, cout_p, 1

            use_halo = a.halo and k == 3 and stride == 1 and pad == 1 and Ww >= 16
            if use_halo:
                pw = torch.empty(lib.cp_packed_halo_weight_bytes(dt, Cout, cin_p), dtype=torch.uint8, device=dev)
                _abi.check(lib.cp_pack_conv3x3_halo_weight(st, dt, w.data_ptr(), Cout, Cin, cin_p, pw.data_ptr()))
            use_gemm = a.gemm and k == 1 and stride == 1 and pad == 0 and Cout >= a.gemm_min_cout and cin_p >= a.gemm_min_k * E
            if use_gemm:
                pw = torch.empty(lib.cp_packed_gemm_weight_bytes(dt, Cout, cin_p), dtype=torch.uint8, device=dev)
                _abi.check(lib.cp_pack_gemm_weight(st, dt, w.data_ptr(), Cout, Cin, cin_p, pw.data_ptr()))
            fnc = lib.cp_conv3x3_halo if use_halo else (lib.cp_gemm_rows if use_gemm else lib.cp_conv2d_igemm)

            def run():
                _abi.check(fnc(st, C.byref(d), x.data_ptr(), pw.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                               res.data_ptr() if res is not None else None, out.data_ptr()))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            fl = 2.0 * B * Ho * Wo * k * k * Cin * Cout
            by = (x.numel() + out.numel() * (2 if has_res else 1)) * es
            print("%-37s %8s %10.1f %10.1f %9.0f" % (name + (" [halo]" if use_halo else (" [gemm]" if use_gemm else "")), "bf16" if dt == CP_BF16 else "fp32", us, fl / us / 1e6, by / us / 1e3))


if __name__ == "__main__":
    main()
