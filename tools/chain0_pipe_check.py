"""Bitwise A/B of two builds of the 64 x 64 x 18 chain (cp_hr_branch_chain, no tail): python tools/chain0_pipe_check.py <libA.so> <libB.so>
Same packed weights, same inputs; outputs must be equal bit for bit; then timing of both at B = 256."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi


def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in _abi.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def timed_pair(runs, rounds=5, reps=20):
    """the two builds alternate inside one process (whichever runs second in a process measures 2-3 % faster): median of `rounds` per build"""
    t = [[], []]
    for r in range(rounds):
        for k in ((0, 1) if r % 2 == 0 else (1, 0)):
            for _ in range(3):
                runs[k]()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                runs[k]()
            e1.record(); torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / reps * 1e3)
    med = [sorted(v)[len(v) // 2] for v in t]
    return "lib 0: median %.1f us (min %.1f)   lib 1: median %.1f us (min %.1f)   ratio %.3f" % (med[0], min(t[0]), med[1], min(t[1]), med[1] / med[0])

libs = [bind(sys.argv[1]), bind(sys.argv[2])]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
Cc, H, W, cp = 18, 64, 64, 24
ok = True
for (B, nsrc, relu_in, seed) in ((3, 1, 0, 1), (5, 3, 1, 2), (2, 4, 1, 3), (256, 3, 1, 4), (256, 1, 1, 5)):
    g = torch.Generator(device=dev).manual_seed(seed)
    shifts = [0, 1, 2, 3][:nsrc]
    srcs = [(torch.randn(B, H >> sh, W >> sh, cp, device=dev, generator=g) * (1.0 if k == 0 else 0.5)).to(torch.bfloat16) for k, sh in enumerate(shifts)]
    for t in srcs:
        t[..., Cc:] = 0                                        # pad channels are exactly zero in the program's tensors
    ws = [(torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * 0.08).contiguous() for _ in range(8)]
    scs = [(torch.rand(Cc, device=dev, generator=g) + 0.5).contiguous() for _ in range(8)]
    n = libs[0].cp_hr_chain_affine_floats(Cc, H, W)
    aff = (torch.randn(8, 2, n, device=dev, generator=g) * 0.1).contiguous()
    aff[..., Cc:] = 0
    outs = []
    for lib in libs:
        blob = torch.zeros(lib.cp_hr_chain_weight_bytes(Cc, H, W), dtype=torch.uint8, device=dev)
        for i in range(8):
            assert lib.cp_pack_hr_chain_weight(st, ws[i].data_ptr(), scs[i].data_ptr(), Cc, H, W, i, blob.data_ptr()) == 0
        out = torch.full((B, H, W, cp), 7.0, device=dev).to(torch.bfloat16)
        arr_p = (C.c_void_p * 4)(*([s.data_ptr() for s in srcs] + [None] * (4 - nsrc)))
        arr_s = (C.c_int32 * 4)(*(shifts + [0] * (4 - nsrc)))
        rc = lib.cp_hr_branch_chain(st, B, Cc, H, W, nsrc, arr_p, arr_s, relu_in, blob.data_ptr(), aff.data_ptr(), out.data_ptr())
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append(out.float().clone())                       # all 24 physical channels: the pads must be exactly zero in both
    same = torch.equal(outs[0], outs[1])
    d = (outs[0] - outs[1]).abs()
    print("B=%3d nsrc=%d relu_in=%d: equal %s  max|d| %.4g  mismatching %d of %d  (|out| max %.3f)" % (B, nsrc, relu_in, same, float(d.max()), int((d > 0).sum()), d.numel(),
                                                                                                float(outs[0].abs().max())), flush=True)
    ok &= same
    if B == 256:
        runs = []
        for lib in libs:
            blob_ = torch.zeros(lib.cp_hr_chain_weight_bytes(Cc, H, W), dtype=torch.uint8, device=dev)
            for i in range(8):
                lib.cp_pack_hr_chain_weight(st, ws[i].data_ptr(), scs[i].data_ptr(), Cc, H, W, i, blob_.data_ptr())
            runs.append((lambda lib=lib, blob_=blob_: lib.cp_hr_branch_chain(st, B, Cc, H, W, nsrc, arr_p, arr_s, relu_in, blob_.data_ptr(), aff.data_ptr(), out.data_ptr())))
        print("   " + timed_pair(runs), flush=True)
# ---- with a tail (cp_hr_branch_chain_tail): the chain output AND every tail output, bit for bit; timing at B = 256
from checkerpose_amd._abi import CpChainTail
for (B, nsrc, tconvs, seed) in ((3, 2, [(36, False)], 11), (5, 3, [(18, True), (72, False)], 12), (4, 4, [(18, True), (18, True), (36, False)], 13),
                                (256, 4, [(18, True), (18, True), (36, False)], 14), (256, 2, [(36, False)], 15)):
    g = torch.Generator(device=dev).manual_seed(seed)
    shifts = [0, 1, 2, 3][:nsrc]
    srcs = [(torch.randn(B, H >> sh, W >> sh, cp, device=dev, generator=g) * (1.0 if k == 0 else 0.5)).to(torch.bfloat16) for k, sh in enumerate(shifts)]
    for t in srcs:
        t[..., Cc:] = 0                                        # pad channels are exactly zero in the program's tensors
    ws = [(torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * 0.08).contiguous() for _ in range(8)]
    scs = [(torch.rand(Cc, device=dev, generator=g) + 0.5).contiguous() for _ in range(8)]
    n = libs[0].cp_hr_chain_affine_floats(Cc, H, W)
    aff = (torch.randn(8, 2, n, device=dev, generator=g) * 0.1).contiguous()
    aff[..., Cc:] = 0
    tws = [((torch.randn(co, Cc, 3, 3, device=dev, generator=g) * 0.1).contiguous(), (torch.rand(co, device=dev, generator=g) + 0.5).contiguous(),
            (torch.randn(co, device=dev, generator=g) * 0.1).contiguous()) for co, _ in tconvs]
    res = []
    for lib in libs:
        blob = torch.zeros(lib.cp_hr_chain_weight_bytes(Cc, H, W), dtype=torch.uint8, device=dev)
        for i in range(8):
            assert lib.cp_pack_hr_chain_weight(st, ws[i].data_ptr(), scs[i].data_ptr(), Cc, H, W, i, blob.data_ptr()) == 0
        tb = torch.zeros(lib.cp_hr_chain_tail_weight_bytes(), dtype=torch.uint8, device=dev)
        tsh = torch.zeros(lib.cp_hr_chain_tail_channels(), dtype=torch.float32, device=dev)
        tl = CpChainTail()
        touts, piece = [], 0
        for i, ((co, relu), (tw_, tsc, tsf)) in enumerate(zip(tconvs, tws)):
            ocp = (co + 7) // 8 * 8
            assert lib.cp_pack_hr_chain_tail_weight(st, tw_.data_ptr(), tsc.data_ptr(), co, piece, ocp, tb.data_ptr()) == 0
            tsh[piece * 8: piece * 8 + co] = tsf
            o = torch.full((B, H // 2, W // 2, ocp), 5.0, device=dev).to(torch.bfloat16)
            tl.out[i], tl.Cout[i], tl.out_cphys[i], tl.relu[i] = o.data_ptr(), co, ocp, 1 if relu else 0
            touts.append(o)
            piece += ocp // 8
        tl.packed_w, tl.shift, tl.nconv = tb.data_ptr(), tsh.data_ptr(), len(tconvs)
        out = torch.full((B, H, W, cp), 7.0, device=dev).to(torch.bfloat16)
        arr_p = (C.c_void_p * 4)(*([s_.data_ptr() for s_ in srcs] + [None] * (4 - nsrc)))
        arr_s = (C.c_int32 * 4)(*(shifts + [0] * (4 - nsrc)))
        # every name the loop rebinds is bound PER ITERATION (round 5 timed libs[1] against itself here: the lambda closed over the globals)
        call = lambda lib=lib, blob=blob, out=out, tl=tl, arr_p=arr_p, arr_s=arr_s: lib.cp_hr_branch_chain_tail(
            st, B, Cc, H, W, nsrc, arr_p, arr_s, 1, blob.data_ptr(), aff.data_ptr(), out.data_ptr(), C.byref(tl))
        assert call() == 0
        torch.cuda.synchronize()
        res.append(([out.float().clone()] + [o.float().clone() for o in touts], call, (blob, tb, tsh, tl, touts, out)))
    same = all(torch.equal(a, b_) for a, b_ in zip(res[0][0], res[1][0]))
    worst = max(float((a - b_).abs().max()) for a, b_ in zip(res[0][0], res[1][0]))
    print("tail B=%3d nsrc=%d convs %s: equal %s  max|d| %.4g" % (B, nsrc, [c for c, _ in tconvs], same, worst), flush=True)
    ok &= same
    if B == 256:
        print("   " + timed_pair([res[0][1], res[1][1]]), flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
