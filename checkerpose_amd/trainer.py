"""Training-side launch program: the forward in train mode (batch-statistics BatchNorm, live
weights re-packed inside the program) followed by its own backward, as ONE static launch list over the C ABI.

Design.  TrainProgram extends engine.Program with a *tape*: every forward emitter that has a gradient pushes a closure
which, when the tape is unwound in reverse, appends the backward launches (BN / bias backward, data-gradient convs --
the forward conv kernels over transformed weights --, MFMA weight-gradients, graph-op marginals).  Activation
gradients live in the storage dtype in per-tensor buffers that are zero-filled once and accumulated into (conv
epilogues accumulate through their `residual` operand), parameter gradients are fp32 slices of ONE flat buffer
(a single bucket for the data-parallel all-reduce, parallel.allreduce_gradients_).  The liveness planner of the base
class keeps every saved activation alive until its last backward reader automatically.  Nothing here computes;
torch supplies memory and the autograd hook (model/_runtime.py: _TrainFn) only.
"""
import ctypes as C
import os
import weakref

import torch

from . import _abi
from ._abi import (CpWgradReduceItem, CpWgradItem, CpFuseBwdItem, CP_WGRAD_ITEM_3X3_S2_SMALL, CP_WGRAD_ITEM_3X3, CP_WGRAD_ITEM_3X3_SMALL, CP_WGRAD_ITEM_GENERIC_BF16,
                   CP_WGRAD_ITEM_GENERIC_F32, ACT_LEAKY, ACT_NONE, ACT_RELU, CP_BF16, CP_F32, CpPackItem, CpWgradDesc, CpBnItem, BN_GROUP_MAX,
                   CP_BN_ITEM_STATS, CP_BN_ITEM_APPLY, CP_BN_ITEM_BWD_SUMS, CP_BN_ITEM_BWD_APPLY)
from .engine import Act, Program, WeightStore, _rup


ALL_TAPS_KINDS = (CP_WGRAD_ITEM_3X3, CP_WGRAD_ITEM_3X3_SMALL, CP_WGRAD_ITEM_3X3_S2_SMALL)      # one block = all nine taps


class TrainWeightStore(WeightStore):
    """Weights change every step: packing launches are appended to the program (right before their consumer) instead
    of being run once at build time; per-channel vectors may be live device tensors passed through unchanged."""

    def __init__(self, lib, sd, dtype, device):
        super().__init__(lib, sd, dtype, device)
        self.prog = None
        self.passthrough = set()
        self.repacks_every_step = True

    def _item(self, rc, item, what):
        _abi.check(rc, what)
        self.prog.add_item("packs", item)

    def pack(self, name, w, Cout, Cin, R, S, cin_phys, cout_rows, transposed=0, phase=0, row_map=None):
        ck = (name, cin_phys, cout_rows, transposed, phase)
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_weight_bytes(self.dtype, cout_rows, cin_phys, R, S), dtype=torch.uint8, device=self.device)
        rm = None
        if row_map is not None:
            rm = torch.tensor(row_map, dtype=torch.int32, device=self.device)
        self.prog.keep += [w, rm, out]
        it = CpPackItem()
        self._item(self.lib.cp_pack_item_conv(self.dtype, w.data_ptr(), Cout, Cin, R, S, cin_phys, transposed, phase,
                                              rm.data_ptr() if rm is not None else None, cout_rows, out.data_ptr(), C.byref(it)),
                   it, "pack:" + name)
        self.cache[ck] = out
        return out

    def pack_halo(self, name, w, Cout, Cin, cin_phys):
        ck = ("halo", name, cin_phys)
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_halo_weight_bytes(self.dtype, Cout, cin_phys), dtype=torch.uint8, device=self.device)
        self.prog.keep += [w, out]
        it = CpPackItem()
        self._item(self.lib.cp_pack_item_halo(self.dtype, w.data_ptr(), Cout, Cin, cin_phys, out.data_ptr(), C.byref(it)), it,
                   "pack_halo:" + name)
        self.cache[ck] = out
        return out

    def pack_gemm(self, name, w, Cout, Cin, cin_phys):
        ck = ("gemm", name, cin_phys)
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_gemm_weight_bytes(self.dtype, Cout, cin_phys), dtype=torch.uint8, device=self.device)
        self.prog.keep += [w, out]
        it = CpPackItem()
        self._item(self.lib.cp_pack_item_gemm(self.dtype, w.data_ptr(), Cout, Cin, cin_phys, out.data_ptr(), C.byref(it)), it,
                   "pack_gemm:" + name)
        self.cache[ck] = out
        return out

    def affine(self, name, scale, shift, rows):
        if id(scale) in self.passthrough and id(shift) in self.passthrough:
            return scale, shift
        return super().affine(name, scale, shift, rows)


_WG_ARENAS = weakref.WeakValueDictionary()      # the TrainPrograms hold the arenas: the last program gone, its 4 GiB go back to torch


def _shared_wgrad_arena(device):
    """The weight-gradient partial-sum arena of `device` + the current stream (CHECKERPOSE_AMD_WGRAD_ARENA_MB, default 4096), and that
    stream's handle: partial sums of two programs that share an arena are ordered only by running on ONE stream, so a program
    remembers the stream it was built for and `_TrainFn` refuses to replay it on another (model/_runtime.py)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    nbytes = int(os.environ.get("CHECKERPOSE_AMD_WGRAD_ARENA_MB", "4096")) << 20
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (idx, stream, nbytes)
    arena = _WG_ARENAS.get(key)
    if arena is None:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _WG_ARENAS[key] = arena
    return arena, stream


class TrainProgram(Program):
    training = True

    def __init__(self, lib, ws, dtype, B, device, pgrad, pslots):
        super().__init__(lib, ws, dtype, B, device)
        ws.prog = self
        self.tape = []
        self.grads = {}            # forward TBuf -> gradient TBuf
        self.pgrad = pgrad         # flat fp32 parameter-gradient buffer
        self.pslots = pslots       # state-dict key -> element offset into pgrad
        self.nograd = set()        # id(TBuf) of tensors that need no gradient (the input image)
        self._bn_ws = {}           # lane -> BatchNorm partial-sum workspace
        # weight-gradient pixel-slice partials: every layer gets its OWN region of this arena and its reduction is DEFERRED; when the
        # arena is full (and at the end of the backward) one cp_wgrad_reduce_batch launch settles every pending layer
        # (ONE arena per device and stream, shared by every TrainProgram -- the stage schedule, a ragged last batch and a second module
        # each build their own program, but the partials never outlive the backward that wrote them and backwards on one stream are
        # ordered)
        self.wg_ws, self.wg_stream = _shared_wgrad_arena(device)
        self._wg_off, self._wg_items, self._wg_keys, self.wg_tabs = 0, [], set(), []
        self.wg_defer = os.environ.get("CHECKERPOSE_AMD_WGRAD_DEFER", "1") != "0"        # A/B: one reduction launch per layer
        # grouped weight gradients: the partial-sum launches of up to wg_group_n layers wait for each other and go out as ONE launch
        # per kernel kind (cp_wgrad_group), every layer cut into its share of wg_group_blocks workgroups instead of a GPU's worth
        self.wg_group = self.wg_defer and os.environ.get("CHECKERPOSE_AMD_WGRAD_GROUP", "1") != "0"
        self.wg_group_n = int(os.environ.get("CHECKERPOSE_AMD_WGRAD_GROUP_N", "64"))
        self.wg_group_flops = float(os.environ.get("CHECKERPOSE_AMD_WGRAD_GROUP_GFLOP", "40")) * 1e9     # bigger layers launch alone
        gb = [int(v) for v in os.environ.get("CHECKERPOSE_AMD_WGRAD_GROUP_BLOCKS", "512,1024,1024").split(",")]
        self.wg_group_blocks = {CP_WGRAD_ITEM_3X3: gb[0], CP_WGRAD_ITEM_3X3_SMALL: gb[1], CP_WGRAD_ITEM_GENERIC_BF16: gb[2],
                                CP_WGRAD_ITEM_GENERIC_F32: gb[2], CP_WGRAD_ITEM_3X3_S2_SMALL: gb[1]}
        self.wg_reduce_n = int(os.environ.get("CHECKERPOSE_AMD_WGRAD_REDUCE_N", "96"))
        self._wgc_pending = []
        # BatchNorm statistics + apply in ONE launch each way (grid barrier in between): measured SLOWER on MI355X -- 42.3 ms per step
        # at best (256 blocks, slow polling) against 35.7 ms for the two-launch forms: the barrier costs >= 10 us per launch -> off
        self.bn_fused = os.environ.get("CHECKERPOSE_AMD_BN_FUSED", "0") == "1"
        # the BatchNorm passes of independent layers (HRNet branches at equal depth, a module's fuse convs) in one launch each
        self.bn_grouped = os.environ.get("CHECKERPOSE_AMD_BN_GROUPED", "1") != "0"
        self.wgrad_flops = {}      # op index -> algorithmic FLOPs of that weight-gradient launch (bench_train roofline)
        self.n_fwd_ops = None
        self._touched, self.pslot_done, self.pslot_done_call = [], {}, {}
        self._ones, self._zeros = {}, {}
        self.acc_total = 0         # fp64 BatchNorm accumulators (elements), one arena zero-filled at the start of a step
        self.acc_arena = None
        self._add(lib.cp_memset_zero, lambda P: (self.acc_arena.data_ptr(), self.acc_arena.numel() * 8), "acc_zero", [], [])
        # weight preparation (packing, data-gradient / EdgeConv views, bias refreshes) reads only parameters: all items of a
        # half are processed by TWO launches at its start (views, then the packs that may consume them) -- cp_pack_batch
        self.items = {("views", 0): [], ("packs", 0): [], ("views", 1): [], ("packs", 1): []}
        self.item_tabs = {}
        self.half = 0
        for kind in ("views", "packs"):
            self._add(lib.cp_pack_batch, (lambda k: lambda P: self._batch_args(k, 0))(kind), "prep_%s_fwd" % kind, [], [])
        self.prep_idx = set()      # op indices of weight-preparation launches
        self.arena = []            # gradient TBufs (not recycled: 288 GB of HBM; one zero fill instead of ~370)
        self.grad_arena = None
        self.kinks = {}            # activation key -> Act whose sign is the (Leaky)ReLU branch taken (tests: oracle FORCE_MASK)
        self.debug = {}            # name -> Acts of interest (tools/train_debug.py with CHECKERPOSE_AMD_NO_RECYCLE=1)

    # ---- lanes: the FORWARD half keeps the eval program's fork/join structure (HRNet branches, decoder || refinement run as
    # parallel hipGraph branches: the latency-bound small-map kernels overlap the 64x64 ones); the backward half is emitted
    # after the last join and stays one ordered stream.  Scratch that is shared between launches is per lane (bn_ws).
    # ---- weight preparation ops (packing, data-gradient / EdgeConv weight views, bias refreshes) read ONLY parameters and
    # write buffers nobody but their one consumer touches: under graph capture they run on a second stream, chunks ahead of
    # the compute stream (run_prep_range), instead of ~1100 serialized 4 us launches on the critical path.
    def add_prep(self, fn, argb, name):
        self.prep_idx.add(len(self.ops))
        self._add(fn, argb, name, [], [])

    def add_item(self, kind, item):
        self.items[(kind, self.half)].append(item)

    def _batch_args(self, kind, half):
        tab = self.item_tabs.get((kind, half))
        if tab is None:
            return (self.dtype, None, None, 0, 0)
        return (self.dtype, tab[0].data_ptr(), tab[1].data_ptr(), tab[2], tab[3])

    # ---- small helpers
    @property
    def bn_ws(self):
        if self.lane not in self._bn_ws:
            self._bn_ws[self.lane] = torch.empty(self.lib.cp_bn_bwd_workspace_bytes(4096), dtype=torch.uint8, device=self.device)
        return self._bn_ws[self.lane]

    def const_vec(self, n, one):
        d = self._ones if one else self._zeros
        n = _rup(n, 16)
        if n not in d:
            d[n] = (torch.ones if one else torch.zeros)(n, dtype=torch.float32, device=self.device)
            self.ws.passthrough.add(id(d[n]))
        return d[n]

    def vec(self, n):
        t = torch.zeros(_rup(n, 16), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        self.ws.passthrough.add(id(t))
        return t

    def pg_ptr(self, key):
        self._touched.append(key)          # the tape closure being unwound writes this parameter's gradient
        return self.pgrad.data_ptr() + 4 * self.pslots[key]

    def memset_t(self, tbuf, name="memset"):
        fn = self.lib.cp_memset_zero
        nb = tbuf.nbytes
        self._add(fn, lambda P: (P(tbuf), nb), name, [], [tbuf])

    def memcpy(self, dst_ptr, src_ptr, nbytes, name="memcpy", prep=False):
        if prep:
            it = CpPackItem()
            _abi.check(self.lib.cp_pack_item_copy_f32(src_ptr, dst_ptr, nbytes // 4, C.byref(it)), name)
            self.add_item("views", it)
        else:
            self._add(self.lib.cp_memcpy_d2d, lambda P: (dst_ptr, src_ptr, nbytes), name, [], [])

    def live_vec(self, n, segments):
        """padded fp32 vector refreshed inside the program from live parameter storage: segments = [(dst_off, tensor,
        src_off, count)]"""
        t = self.vec(n)
        for doff, src, soff, cnt in segments:
            self.keep.append(src)
            self.memcpy(t.data_ptr() + 4 * doff, src.data_ptr() + 4 * soff, 4 * cnt, "refresh_vec", prep=True)
        return t

    def scratch_f32(self, numel):
        t = torch.empty(numel, dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def grad_of(self, a: Act):
        """gradient view matching `a` (same geometry) -- buffer created and zero-filled on first request"""
        g = self.grads.get(a.tbuf)
        if g is None:
            g = self.tensor(a.tbuf.nbytes, es=1)
            g.nbytes = a.tbuf.nbytes
            self.grads[a.tbuf] = g
            self.arena.append(g)       # gradient buffers live in one arena zero-filled by ONE launch (finalize())
        return Act(g, a.B, a.H, a.W, a.C, a.Cphys, a.cstride, a.coff)

    def needs_grad(self, a: Act):
        return id(a.tbuf) not in self.nograd

    # ---- train-mode BatchNorm: two launches per pass (column sums by fp64 atomics into a per-layer accumulator pair, then the
    # consumer kernel derives the coefficients itself) -- the accumulators of the whole step are zero-filled by ONE launch
    def _acc_slot(self, C_):
        off = self.acc_total
        self.acc_total += int(self.lib.cp_bn_acc_doubles(C_)) + 2      # + the fused launches' barrier counter (zeroed with the sums)
        return off

    def _ctr_ptr(self, off, C_):
        return self.acc_arena.data_ptr() + 8 * (off + int(self.lib.cp_bn_acc_doubles(C_)))

    def _acc_ptr(self, off):
        return self.acc_arena.data_ptr() + 8 * off

    def bn_stats(self, x: Act, C_, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
        bn = dict(mean=self.vec(C_), rstd=self.vec(C_), gamma=gamma, beta=beta, rmean=rmean, rvar=rvar, C=C_, acc=self._acc_slot(C_),
                  momentum=momentum, eps=eps)
        self.keep += [gamma, beta, rmean, rvar]
        xt = x.tbuf
        M = x.B * x.H * x.W
        off = bn["acc"]
        if self.bn_fused:                 # statistics + apply in one launch: emitted by bn_apply (same tensor, directly behind)
            bn["stats_of"] = x
            return bn
        self._add(self.lib.cp_bn_stats_accumulate, lambda P: (self.dtype, P(xt), M, C_, x.cstride, x.coff, self._acc_ptr(off)),
                  "bn_stats", [xt], [])
        return bn

    def bn_apply(self, x: Act, bn, residual, out: Act, act, slope=0.0):
        xt, ot = x.tbuf, out.tbuf
        rt = residual.tbuf if residual is not None else None
        M = x.B * x.H * x.W
        rcs, rco = (residual.cstride, residual.coff) if residual is not None else (0, 0)
        off = bn["acc"]
        a1 = (bn["gamma"].data_ptr(), bn["beta"].data_ptr(), bn["rmean"].data_ptr(), bn["rvar"].data_ptr(), bn["momentum"], bn["eps"])
        mp, rp = bn["mean"].data_ptr(), bn["rstd"].data_ptr()
        if bn.pop("stats_of", None) is x:
            Cc = bn["C"]
            self._add(self.lib.cp_bn_train_fused,
                      lambda P: (self.dtype, P(xt), x.cstride, x.coff, self._acc_ptr(off), self._ctr_ptr(off, Cc)) + a1 +
                                (P(rt) if rt is not None else None, rcs, rco, P(ot), out.cstride, out.coff, M, x.C, act, slope, mp, rp),
                      "bn_fused", [xt, rt], [ot])
            return out
        self._add(self.lib.cp_bn_apply,
                  lambda P: (self.dtype, P(xt), x.cstride, x.coff, self._acc_ptr(off)) + a1 +
                            (P(rt) if rt is not None else None, rcs, rco, P(ot), out.cstride, out.coff, M, x.C, act, slope, mp, rp),
                  "bn_apply", [xt, rt], [ot])
        return out

    def affine_act(self, x: Act, scale, shift, residual, out: Act, act, slope=0.0):
        xt, ot = x.tbuf, out.tbuf
        rt = residual.tbuf if residual is not None else None
        M = x.B * x.H * x.W
        sp, tp = scale.data_ptr(), shift.data_ptr()
        rcs, rco = (residual.cstride, residual.coff) if residual is not None else (0, 0)
        self._add(self.lib.cp_affine_act,
                  lambda P: (self.dtype, P(xt), x.cstride, x.coff, sp, tp, P(rt) if rt is not None else None, rcs, rco, P(ot),
                             out.cstride, out.coff, M, x.C, act, slope), "affine_act", [xt, rt], [ot])
        return out

    def bn_bwd(self, gy: Act, y: Act, raw: Act, bn, act, slope, gres: Act, dgamma_ptr, dbeta_ptr):
        """in place on gy: gy <- d loss / d raw ; gres += dz ; dgamma / dbeta written.  raw None = bias-only layer."""
        gt = gy.tbuf
        yt = y.tbuf if (y is not None and act != ACT_NONE) else None
        rt = raw.tbuf if raw is not None else None
        grt = gres.tbuf if gres is not None else None
        M = gy.B * gy.H * gy.W
        C_ = gy.C
        mean_p = bn["mean"].data_ptr() if bn else None
        rstd_p = bn["rstd"].data_ptr() if bn else None
        gam_p = bn["gamma"].data_ptr() if bn else None
        ycs, yco = (y.cstride, y.coff) if yt is not None else (0, 0)
        rcs, rco = (raw.cstride, raw.coff) if raw is not None else (0, 0)
        gcs, gco = (gres.cstride, gres.coff) if gres is not None else (0, 0)
        off = self._acc_slot(C_)
        if self.bn_fused:
            self._add(self.lib.cp_bn_bwd_fused,
                      lambda P: (self.dtype, P(gt), gy.cstride, gy.coff, P(yt) if yt is not None else None, ycs, yco,
                                 P(rt) if rt is not None else None, rcs, rco, mean_p, rstd_p, gam_p, self._acc_ptr(off), self._ctr_ptr(off, C_),
                                 M, C_, act, slope, P(gt), gy.cstride, gy.coff, P(grt) if grt is not None else None, gcs, gco, 1,
                                 dgamma_ptr, dbeta_ptr),
                      "bn_bwd_fused", [gt, yt, rt, grt], [gt, grt])
            return
        self._add(self.lib.cp_bn_bwd_accumulate,
                  lambda P: (self.dtype, P(gt), gy.cstride, gy.coff, P(yt) if yt is not None else None, ycs, yco,
                             P(rt) if rt is not None else None, rcs, rco, mean_p, rstd_p, M, C_, act, slope, self._acc_ptr(off)),
                  "bn_bwd_acc", [gt, yt, rt], [])
        self._add(self.lib.cp_bn_bwd_apply,
                  lambda P: (self.dtype, P(gt), gy.cstride, gy.coff, P(yt) if yt is not None else None, ycs, yco,
                             P(rt) if rt is not None else None, rcs, rco, mean_p, rstd_p, gam_p, self._acc_ptr(off), M, C_, act, slope,
                             P(gt), gy.cstride, gy.coff, P(grt) if grt is not None else None, gcs, gco, 1, dgamma_ptr, dbeta_ptr),
                  "bn_bwd", [gt, yt, rt, grt], [gt, grt])

    # ---- grouped BatchNorm passes: the same pass of INDEPENDENT layers (the branches of an HRNet module at equal depth, the fuse
    # convs of a module) in ONE launch (cp_bn_group): at B = 32 a single pass is a 5-13 us launch over 0.3-5 MB, latency not bandwidth
    def _bn_group(self, kind, builders, name, reads, writes):
        """builders: callables P -> CpBnItem, run when the workspace is planned (pointers are final then)"""
        def argb(P):
            items = [b(P) for b in builders]
            raw, prefix, total = _abi.device_table(items, [it.blocks for it in items], self.device)
            self.keep += [raw, prefix]
            return (self.dtype, kind, raw.data_ptr(), prefix.data_ptr(), len(items), total, max(int(it.lds_bytes) for it in items))
        self._add(self.lib.cp_bn_group, argb, name, reads, writes)

    def bn_train_group(self, members):
        """members: [(raw Act, C, gamma, beta, rmean, rvar, residual Act | None, out Act, act, slope)] of independent layers ->
        [bn dict]; statistics of all of them in one launch, apply (+residual, +activation) in a second one"""
        if len(members) == 1 or self.bn_fused or not self.bn_grouped:
            out = []
            for raw, C_, g, b, rm, rv, res, y, act, slope in members:
                bn = self.bn_stats(raw, C_, g, b, rm, rv)
                self.bn_apply(raw, bn, res, y, act, slope)
                out.append(bn)
            return out
        bns, sb, ab, reads_s, reads_a, writes_a = [], [], [], [], [], []
        lib, dt = self.lib, self.dtype
        for raw, C_, g, b, rm, rv, res, y, act, slope in members:
            bn = dict(mean=self.vec(C_), rstd=self.vec(C_), gamma=g, beta=b, rmean=rm, rvar=rv, C=C_, acc=self._acc_slot(C_),
                      momentum=0.1, eps=1e-5)
            self.keep += [g, b, rm, rv]
            bns.append(bn)
            M = raw.B * raw.H * raw.W

            def stats(P, raw=raw, C_=C_, M=M, off=bn["acc"]):
                it = CpBnItem()
                _abi.check(lib.cp_bn_item_stats(dt, P(raw.tbuf), M, C_, raw.cstride, raw.coff, self._acc_ptr(off), C.byref(it)), "cp_bn_item_stats")
                return it

            def apply_(P, raw=raw, M=M, bn=bn, res=res, y=y, act=act, slope=slope):
                it = CpBnItem()
                rp, rcs, rco = (P(res.tbuf), res.cstride, res.coff) if res is not None else (None, 0, 0)
                _abi.check(lib.cp_bn_item_apply(dt, P(raw.tbuf), raw.cstride, raw.coff, self._acc_ptr(bn["acc"]), bn["gamma"].data_ptr(),
                                                bn["beta"].data_ptr(), bn["rmean"].data_ptr(), bn["rvar"].data_ptr(), bn["momentum"],
                                                bn["eps"], rp, rcs, rco, P(y.tbuf), y.cstride, y.coff, M, raw.C, act, slope,
                                                bn["mean"].data_ptr(), bn["rstd"].data_ptr(), C.byref(it)), "cp_bn_item_apply")
                return it
            sb.append(stats)
            ab.append(apply_)
            reads_s.append(raw.tbuf)
            reads_a += [raw.tbuf] + ([res.tbuf] if res is not None else [])
            writes_a.append(y.tbuf)
        for lo in range(0, len(members), BN_GROUP_MAX):
            hi = lo + BN_GROUP_MAX
            self._bn_group(CP_BN_ITEM_STATS, sb[lo:hi], "bn_stats_group:%d" % len(sb[lo:hi]), reads_s, [])
        for lo in range(0, len(members), BN_GROUP_MAX):
            hi = lo + BN_GROUP_MAX
            self._bn_group(CP_BN_ITEM_APPLY, ab[lo:hi], "bn_apply_group:%d" % len(ab[lo:hi]), reads_a, writes_a)
        return bns

    def bn_bwd_group(self, members):
        """members: [(gy, y, raw, bn, act, slope, gres, dgamma_ptr, dbeta_ptr)] as bn_bwd's arguments, independent layers"""
        if len(members) == 1 or self.bn_fused or not self.bn_grouped:
            for m in members:
                self.bn_bwd(*m)
            return
        lib, dt = self.lib, self.dtype
        sb, ab, reads_s, reads_a, writes_a = [], [], [], [], []
        for gy, y, raw, bn, act, slope, gres, dgp, dbp in members:
            use_y = y is not None and act != ACT_NONE
            M, C_ = gy.B * gy.H * gy.W, gy.C
            off = self._acc_slot(C_)
            mean_p = bn["mean"].data_ptr() if bn else None
            rstd_p = bn["rstd"].data_ptr() if bn else None
            gam_p = bn["gamma"].data_ptr() if bn else None

            def ptrs(P, gy=gy, y=y, raw=raw, use_y=use_y):
                yv = (P(y.tbuf), y.cstride, y.coff) if use_y else (None, 0, 0)
                xv = (P(raw.tbuf), raw.cstride, raw.coff) if raw is not None else (None, 0, 0)
                return (dt, P(gy.tbuf), gy.cstride, gy.coff) + yv + xv

            def sums(P, ptrs=ptrs, M=M, C_=C_, act=act, slope=slope, off=off, mean_p=mean_p, rstd_p=rstd_p):
                it = CpBnItem()
                _abi.check(lib.cp_bn_item_bwd_sums(*(ptrs(P) + (mean_p, rstd_p, M, C_, act, slope, self._acc_ptr(off), C.byref(it)))),
                           "cp_bn_item_bwd_sums")
                return it

            def apply_(P, ptrs=ptrs, gy=gy, gres=gres, M=M, C_=C_, act=act, slope=slope, off=off, mean_p=mean_p, rstd_p=rstd_p,
                       gam_p=gam_p, dgp=dgp, dbp=dbp):
                it = CpBnItem()
                gr = (P(gres.tbuf), gres.cstride, gres.coff) if gres is not None else (None, 0, 0)
                _abi.check(lib.cp_bn_item_bwd_apply(*(ptrs(P) + (mean_p, rstd_p, gam_p, self._acc_ptr(off), M, C_, act, slope,
                                                                 P(gy.tbuf), gy.cstride, gy.coff) + gr + (1, dgp, dbp, C.byref(it)))),
                           "cp_bn_item_bwd_apply")
                return it
            sb.append(sums)
            ab.append(apply_)
            rd = [gy.tbuf] + ([y.tbuf] if use_y else []) + ([raw.tbuf] if raw is not None else [])
            reads_s += rd
            reads_a += rd + ([gres.tbuf] if gres is not None else [])
            writes_a += [gy.tbuf] + ([gres.tbuf] if gres is not None else [])
        for lo in range(0, len(members), BN_GROUP_MAX):
            hi = lo + BN_GROUP_MAX
            self._bn_group(CP_BN_ITEM_BWD_SUMS, sb[lo:hi], "bn_bwd_acc_group:%d" % len(sb[lo:hi]), reads_s, [])
        for lo in range(0, len(members), BN_GROUP_MAX):
            hi = lo + BN_GROUP_MAX
            self._bn_group(CP_BN_ITEM_BWD_APPLY, ab[lo:hi], "bn_bwd_group:%d" % len(ab[lo:hi]), reads_a, writes_a)

    # ---- dense layer backward
    def wgrad(self, dy: Act, x: Act, dw_ptr, Cout, Cin, R, S, stride, pad, Ho=None, Wo=None, base=0, sco=None, sci=None):
        d = CpWgradDesc()
        d.dtype, d.B, d.H, d.W = self.dtype, x.B, x.H, x.W
        d.Ho, d.Wo = (dy.H, dy.W) if Ho is None else (Ho, Wo)
        d.Cout, d.dy_cstride, d.dy_coff = Cout, dy.cstride, dy.coff
        d.Cin, d.x_cstride, d.x_coff = Cin, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad = R, S, stride, pad
        d.dw_base, d.dw_sco, d.dw_sci, d.dw_sr, d.dw_ss = base, (Cin * R * S if sco is None else sco), (R * S if sci is None else sci), S, 1
        self.keep.append(d)
        dref = C.byref(d)
        dt, xt = dy.tbuf, x.tbuf
        name = "wgrad:%d->%d k%d s%d %dx%d" % (Cin, Cout, R, stride, x.H, x.W)
        flops = 2 * dy.B * d.Ho * d.Wo * R * S * Cin * Cout                # algorithmic, unpadded
        if self.wg_group and flops < self.wg_group_flops:
            self._wgc_pending.append(dict(d=d, dref=dref, dt=dt, xt=xt, dw=dw_ptr, flops=flops, k3s1=(R == 3 and stride == 1),
                                          pix=dy.B * d.Ho * d.Wo, cc=((Cout + 63) // 64) * ((Cin + 63) // 64), taps=R * S))
            if len(self._wgc_pending) >= self.wg_group_n:
                self.flush_wgrad_compute()
            return
        if not self.wg_defer:
            wsp, wsn = self.wg_ws.data_ptr(), min(self.wg_ws.numel(), 160 << 20)
            self._add(self.lib.cp_conv2d_wgrad_ws, lambda P: (dref, P(dt), P(xt), dw_ptr, wsp, wsn), name, [dt, xt], [])
        else:
            need = (int(self.lib.cp_conv2d_wgrad_scratch_bytes(dref)) + 255) // 256 * 256
            if self._wg_off + need > self.wg_ws.numel():
                self.flush_wgrad_compute()
                self._flush_wgrad_reduce()
            need = min(need, self.wg_ws.numel())
            wsp = self.wg_ws.data_ptr() + self._wg_off
            item, spare = CpWgradReduceItem(), CpWgradReduceItem()
            base = self.wg_ws.data_ptr()      # any aligned non-null pointer: the plan only validates dy / x, it launches nothing
            _abi.check(self.lib.cp_conv2d_wgrad_plan(dref, base, base, dw_ptr, wsp, need, C.byref(item)), "cp_conv2d_wgrad_plan")
            self.keep.append(spare)
            sref = C.byref(spare)
            self._add(self.lib.cp_conv2d_wgrad_deferred, lambda P: (dref, P(dt), P(xt), dw_ptr, wsp, need, sref), name, [dt, xt], [])
            if item.ws:                       # this layer owes a reduction (tiny layers add with atomics instead)
                self._wg_items.append(item)
                self._wg_off += need
        self.wgrad_flops[len(self.ops) - 1] = flops

    def flush_wgrad_compute(self):
        """the partial-sum launches of the waiting layers: ONE launch per kernel kind.  A layer's share of the launch's workgroups is
        proportional to its work (pixels x channel blocks), so all blocks of a launch run about equally long."""
        pend, self._wgc_pending = self._wgc_pending, []
        if not pend:
            return
        lib, arena, asz = self.lib, self.wg_ws.data_ptr(), self.wg_ws.numel()
        for m in pend:                        # kernel kind of every layer (depends on the descriptor only)
            ci, ri = CpWgradItem(), CpWgradReduceItem()
            _abi.check(lib.cp_conv2d_wgrad_item(m["dref"], arena, arena, m["dw"], arena, asz, 0, C.byref(ci), C.byref(ri)), "cp_conv2d_wgrad_item")
            m["kind"] = int(ci.kind)
            m["work"] = m["pix"] * m["cc"] * (1 if ci.kind in ALL_TAPS_KINDS else m["taps"])
        kinds = sorted({m["kind"] for m in pend})
        for kind in kinds:                    # every layer's share of its launch, and the partial tiles that share needs
            mem = [m for m in pend if m["kind"] == kind]
            total = float(sum(m["work"] for m in mem))
            for m in mem:
                m["target"] = max(1, int(round(self.wg_group_blocks[kind] * m["work"] / total)))
                ci, ri = CpWgradItem(), CpWgradReduceItem()
                _abi.check(lib.cp_conv2d_wgrad_item(m["dref"], arena, arena, m["dw"], arena, asz, m["target"], C.byref(ci), C.byref(ri)), "cp_conv2d_wgrad_item")
                m["need"] = (int(ri.S) * int(ri.GY) * int(ri.taps_in_block) * 4096 * 4 + 255) // 256 * 256 if ri.ws else 0
                m["blocks"] = int(ci.blocks)
        need_all = sum(m["need"] for m in pend)
        if need_all > asz:
            raise RuntimeError("checkerpose_amd: the weight-gradient arena (%d MB, CHECKERPOSE_AMD_WGRAD_ARENA_MB) is smaller than the partial "
                               "tiles of one group of layers (%d MB)" % (asz >> 20, (need_all >> 20) + 1))
        if self._wg_off + need_all > asz:     # the reductions owed so far (their producers are all emitted) free the arena
            self._flush_wgrad_reduce()
        for kind in kinds:
            mem = [m for m in pend if m["kind"] == kind]
            builders = []
            for m in mem:
                if m["need"]:
                    m["wsp"], m["wsn"] = arena + self._wg_off, m["need"]
                    ci, ri = CpWgradItem(), CpWgradReduceItem()
                    _abi.check(lib.cp_conv2d_wgrad_item(m["dref"], arena, arena, m["dw"], m["wsp"], m["wsn"], m["target"], C.byref(ci), C.byref(ri)),
                               "cp_conv2d_wgrad_item")
                    assert ri.ws == m["wsp"] and int(ci.blocks) == m["blocks"]
                    self._wg_items.append(ri)
                    self._wg_off += m["need"]
                else:
                    m["wsp"], m["wsn"] = arena, asz

                def build(P, m=m):
                    it, spare = CpWgradItem(), CpWgradReduceItem()
                    _abi.check(lib.cp_conv2d_wgrad_item(m["dref"], P(m["dt"]), P(m["xt"]), m["dw"], m["wsp"], m["wsn"], m["target"], C.byref(it),
                                                        C.byref(spare)), "cp_conv2d_wgrad_item")
                    assert int(it.blocks) == m["blocks"] and int(it.kind) == m["kind"]
                    return it
                builders.append(build)

            def argb(P, builders=builders, kind=kind):
                items = [b(P) for b in builders]
                raw, prefix, total = _abi.device_table(items, [it.blocks for it in items], self.device)
                self.keep += [raw, prefix]
                return (kind, raw.data_ptr(), prefix.data_ptr(), len(items), total)
            k3 = all(m["k3s1"] for m in mem)
            self._add(lib.cp_wgrad_group, argb, "wgrad_group:%s x%d" % (" k3 s1 " if k3 else "mixed", len(mem)),
                      [t for m in mem for t in (m["dt"], m["xt"])], [])
            self.wgrad_flops[len(self.ops) - 1] = sum(m["flops"] for m in mem)
        if len(self._wg_items) >= self.wg_reduce_n:        # settle in a few instalments: the data-parallel step's gradient buckets
            self.flush_wgrad()                              # complete (and start their all-reduce) while the backward goes on

    def _flush_wgrad_reduce(self):
        if self._wg_items:
            items = self._wg_items
            raw, prefix, acc = _abi.device_table(items, [self.lib.cp_wgrad_reduce_item_blocks(C.byref(it)) for it in items], self.device)
            self.wg_tabs.append((raw, prefix))
            n = len(items)
            self._add(self.lib.cp_wgrad_reduce_batch, lambda P: (raw.data_ptr(), prefix.data_ptr(), n, acc), "wgrad_reduce_batch:%d" % n, [], [])
        self._wg_off, self._wg_items = 0, []

    def flush_wgrad(self):
        """every weight-gradient launch still waiting, then one launch for every reduction owed so far; the parameters touched since
        the last flush are final only behind it (gradient buckets of the data-parallel step, plan_gradient_buckets)"""
        self.flush_wgrad_compute()
        self._flush_wgrad_reduce()
        for k in self._wg_keys:
            self.pslot_done[k] = len(self.ops)
        self._wg_keys = set()

    def weight_dgrad(self, w, Cout, Cin, R, S):
        wt = self.scratch_f32(Cout * Cin * R * S).view(Cin, Cout, R, S)
        self.keep.append(w)
        it = CpPackItem()
        _abi.check(self.lib.cp_pack_item_dgrad_view(w.data_ptr(), Cout, Cin, R, S, wt.data_ptr(), C.byref(it)), "weight_dgrad")
        self.add_item("views", it)
        return wt

    def conv_backward(self, key, w, x: Act, g: Act, R, S, stride, pad):
        """g = d loss / d (raw conv output) in channels-last; accumulates into grad(x), writes the weight gradient."""
        Cout, Cin = w.shape[0], w.shape[1]
        self.wgrad(g, x, self.pg_ptr(key + ".weight"), Cout, Cin, R, S, stride, pad)
        if self.needs_grad(x):
            self._dgrad(key, w, x, g, R, S, stride, pad)

    def conv_backward_group(self, members):
        """conv_backward of INDEPENDENT layers [(key, w, x, g, R, S, stride, pad)]: the 3x3 / stride 1 data-gradient convs go out as
        grouped launches (engine.conv3x3_group; members that accumulate into the same input gradient in separate rounds)"""
        dg = []
        for key, w, x, g, R, S, stride, pad in members:
            Cout, Cin = w.shape[0], w.shape[1]
            self.wgrad(g, x, self.pg_ptr(key + ".weight"), Cout, Cin, R, S, stride, pad)
            if not self.needs_grad(x):
                continue
            if stride == 1 and R == 3 and S == 3 and pad == 1:
                gx = self.grad_of(x)
                wt = self.weight_dgrad(w, Cout, Cin, 3, 3)
                dg.append((g, key + "#dgrad", wt, self.const_vec(Cin, True), self.const_vec(Cin, False), ACT_NONE, 0.0, gx, gx))
            else:
                self._dgrad(key, w, x, g, R, S, stride, pad)
        while dg:
            seen, now, later = set(), [], []
            for m in dg:
                (later if id(m[8].tbuf) in seen else now).append(m)
                seen.add(id(m[8].tbuf))
            self.conv3x3_group(now)
            dg = later

    def _dgrad(self, key, w, x: Act, g: Act, R, S, stride, pad):
        Cout, Cin = w.shape[0], w.shape[1]
        gx = self.grad_of(x)
        one, zero = self.const_vec(Cin, True), self.const_vec(Cin, False)
        if stride == 1:
            wt = self.weight_dgrad(w, Cout, Cin, R, S)
            self.conv(g, key + "#dgrad", wt, one, zero, R, S, 1, R - 1 - pad, Cin, residual=gx, out=gx)
        elif stride == 2 and R == 3 and S == 3 and pad == 1 and x.H == 2 * g.H and x.W == 2 * g.W:
            for ph in range(4):       # ConvTranspose2d(k3,s2,p1,op1) phases over w read as (Cin_t = Cout, Cout_t = Cin, 3, 3)
                a, b = ph >> 1, ph & 1
                self.conv(g, key + "#dgrad", w, one, zero, 1 + a, 1 + b, 1, 0, gx.Cphys, transposed=1, phase=ph, residual=gx,
                          ostr=(gx.coff + (a * gx.W + b) * gx.cstride, gx.H * gx.W * gx.cstride, 2 * gx.W * gx.cstride,
                                2 * gx.cstride, 1), out_tbuf=gx.tbuf, out_hw=(g.H, g.W))
        elif stride == 2 and R == 1 and S == 1 and pad == 0:
            wt = self.weight_dgrad(w, Cout, Cin, 1, 1)
            self.conv(g, key + "#dgrad", wt, one, zero, 1, 1, 1, 0, gx.Cphys, residual=gx,
                      ostr=(gx.coff, gx.H * gx.W * gx.cstride, 2 * gx.W * gx.cstride, 2 * gx.cstride, 1), out_tbuf=gx.tbuf,
                      out_hw=(g.H, g.W))
        else:
            raise RuntimeError("no data-gradient path for conv %s (k=%d, stride=%d, pad=%d)" % (key, R, stride, pad))

    def upsample2x_bwd(self, gout: Act, gin: Act):
        gt, it = gout.tbuf, gin.tbuf
        tail = (gin.B, gin.H, gin.W, gin.Cphys, gout.cstride, gout.coff, gin.cstride, gin.coff, 1)
        self._add(self.lib.cp_upsample2x_bilinear_ac_bwd, lambda P: (self.dtype, P(gt), P(it)) + tail, "upsample2x_bwd", [gt, it], [it])

    def fuse_sum_bwd(self, gout: Act, out: Act, gsrc: Act, shift, relu):
        gt, ot, st_ = gout.tbuf, out.tbuf, gsrc.tbuf
        tail = (gsrc.B, gsrc.H, gsrc.W, gsrc.Cphys, int(shift), 1 if relu else 0, 1)
        self._add(self.lib.cp_fuse_sum_act_bwd, lambda P: (self.dtype, P(gt), P(ot), P(st_)) + tail, "fuse_sum_bwd", [gt, ot, st_], [st_])

    def fuse_sum_bwd_group(self, members):
        """members: [(gout, out, gsrc, shift, relu)] writing DISTINCT gradient tensors (a module's whole fuse layer): one launch"""
        if len(members) == 1 or not self.bn_grouped:
            for m in members:
                self.fuse_sum_bwd(*m)
            return
        lib, dt = self.lib, self.dtype
        for lo in range(0, len(members), 64):
            mem = members[lo:lo + 64]

            def argb(P, mem=mem):
                items, nbs = [], []
                for gout, out, gsrc, shift, relu in mem:
                    it, nb = CpFuseBwdItem(), C.c_uint32()
                    _abi.check(lib.cp_fuse_sum_act_bwd_item(dt, P(gout.tbuf), P(out.tbuf), P(gsrc.tbuf), gsrc.B, gsrc.H, gsrc.W, gsrc.Cphys,
                                                            int(shift), 1 if relu else 0, 1, C.byref(it), C.byref(nb)), "cp_fuse_sum_act_bwd_item")
                    items.append(it)
                    nbs.append(nb.value)
                raw, prefix, total = _abi.device_table(items, nbs, self.device)
                self.keep += [raw, prefix]
                return (dt, raw.data_ptr(), prefix.data_ptr(), len(items), total)
            tbs = [t for gout, out, gsrc, _, _ in mem for t in (gout.tbuf, out.tbuf, gsrc.tbuf)]
            self._add(lib.cp_fuse_sum_act_bwd_group, argb, "fuse_sum_bwd_group:%d" % len(mem), tbs, [m[2].tbuf for m in mem])

    def maxpool_bwd(self, x: Act, gout: Act, gin: Act):
        xt, gt, it = x.tbuf, gout.tbuf, gin.tbuf
        assert x.coff == 0 and x.cstride == x.Cphys
        self._add(self.lib.cp_maxpool3x3s2_bwd, lambda P: (self.dtype, P(xt), P(gt), P(it), x.B, x.H, x.W, x.Cphys, 1),
                  "maxpool_bwd", [xt, gt, it], [it])

    def strided_to_act(self, src_ptr_fn, src_dtype, base, sb, sp, sc, out: Act, C_, reads=()):
        """strided tensor -> dense channels-last Act (out must be dense: coff 0, cstride == Cphys)"""
        assert out.coff == 0 and out.cstride == out.Cphys
        ot = out.tbuf
        HW = out.H * out.W
        self._add(self.lib.cp_strided_to_nhwc,
                  lambda P: (self.dtype, src_ptr_fn(P), src_dtype, base, sb, sp, sc, P(ot), out.B, HW, C_, out.Cphys),
                  "strided_to_nhwc", list(reads), [ot])

    # ---- train-mode EdgeConv
    def edge_weight_view(self, w, Co, Ci, mode):
        out = self.scratch_f32(2 * Co * Ci)
        self.keep.append(w)
        it = CpPackItem()
        _abi.check(self.lib.cp_pack_item_edge_view(w.data_ptr(), Co, Ci, mode, out.data_ptr(), C.byref(it)), "edge_weight_view")
        self.add_item("views", it)
        return out.view(2 * Co, Ci, 1, 1) if mode == 0 else out.view(Ci, 2 * Co, 1, 1)

    def edge_train_fwd(self, pq: Act, graph, gamma, beta, rmean, rvar, out: Act, Co, slope, momentum=0.1, eps=1e-5):
        N = pq.W
        st_ = dict(scale=self.vec(Co), shift=self.vec(Co), mean=self.vec(Co), rstd=self.vec(Co), gamma=gamma,
                   kstar=torch.empty(pq.B * N * Co, dtype=torch.uint8, device=self.device),
                   ws=torch.empty(self.lib.cp_edge_train_workspace_bytes(pq.B, Co), dtype=torch.uint8, device=self.device))
        self.keep += [st_, gamma, beta, rmean, rvar]
        pt, ot = pq.tbuf, out.tbuf
        gp = graph["gids"].data_ptr() if graph["gids"] is not None else None
        a1 = (graph["idx"].data_ptr(), gp, gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), momentum, eps)
        a2 = (st_["kstar"].data_ptr(), st_["scale"].data_ptr(), st_["shift"].data_ptr(), st_["mean"].data_ptr(),
              st_["rstd"].data_ptr(), st_["ws"].data_ptr(), pq.B, N, graph["K"], Co, graph["G"], slope)
        self._add(self.lib.cp_edgeconv_train_fwd, lambda P: (self.dtype, P(pt)) + a1 + (P(ot), out.cstride, out.coff) + a2,
                  "edge_train_fwd", [pt], [ot])
        return st_

    def edge_train_bwd(self, pq: Act, graph, st_, out: Act, gout: Act, D: Act, dgamma_ptr, dbeta_ptr, Co, slope):
        N = pq.W
        pt, ot, gt, dt = pq.tbuf, out.tbuf, gout.tbuf, D.tbuf
        gp = graph["gids"].data_ptr() if graph["gids"] is not None else None
        a1 = (graph["idx"].data_ptr(), graph["rev_ptr"].data_ptr(), graph["rev_edge"].data_ptr(), gp)
        a3 = (st_["gamma"].data_ptr(), st_["mean"].data_ptr(), st_["rstd"].data_ptr())
        a4 = (dgamma_ptr, dbeta_ptr, st_["ws"].data_ptr(), pq.B, N, graph["K"], Co, graph["G"], slope)
        kp = st_["kstar"].data_ptr()
        self._add(self.lib.cp_edgeconv_train_bwd,
                  lambda P: (self.dtype, P(pt)) + a1 + (P(ot), out.cstride, out.coff, kp, P(gt), gout.cstride, gout.coff) + a3 +
                            (P(dt),) + a4, "edge_train_bwd", [pt, ot, gt], [dt])

    def index2feat_bwd(self, gout: Act, xid_t, yid_t, mask_t, dpatch_f32, N, Hp, Wp, E_ch, k):
        gt = gout.tbuf
        args = (xid_t.data_ptr(), yid_t.data_ptr(), mask_t.data_ptr(), dpatch_f32.data_ptr(), gout.B, N, Hp, Wp, E_ch, k,
                gout.cstride, gout.coff)
        self._add(self.lib.cp_index2feat_gather_bwd_t, lambda P: (self.dtype, P(gt)) + args, "index2feat_bwd", [gt], [])

    def finalize(self):
        for key, items in self.items.items():
            if not items:
                continue
            raw, prefix, acc = _abi.device_table(items, [(int(it.total) + 255) // 256 for it in items], self.device)
            self.item_tabs[key] = (raw, prefix, len(items), acc)
        self.acc_arena = torch.zeros(max(self.acc_total, 2), dtype=torch.float64, device=self.device)
        total = sum(t.nbytes for t in self.arena)
        self.grad_arena = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)
        off = 0
        for t in self.arena:
            t.fixed = self.grad_arena[off:off + t.nbytes]
            off += t.nbytes
        super().finalize()
        # n_fwd_ops was recorded in units of emitted ops (launches + fork/sync/join markers): convert to launches
        markers = ("__fork__", "__sync__", "__join__", "__mark__", "__wait__")
        nmark, before = 0, []
        for op in self.ops:                     # before[i] = launches among ops[:i]
            before.append(nmark)
            nmark += 0 if op[0] in markers else 1
        before.append(nmark)
        self.pslot_done_call = {k: before[i] for k, i in self.pslot_done.items()}
        self.n_fwd_ops = sum(1 for op in self.ops[:self.n_fwd_ops] if op[0] not in markers)
        ci, prep_calls, flops = 0, set(), {}
        for i, op in enumerate(self.ops):
            if op[0] in markers:
                continue
            if i in self.prep_idx:
                prep_calls.add(ci)
            if i in self.wgrad_flops:
                flops[ci] = self.wgrad_flops[i]
            ci += 1
        self.prep_calls = prep_calls
        self.wgrad_flops = flops           # re-keyed by launch (call) index, like everything bench_train reads
        return self

    def zero_grad_arena(self):
        """ONE zero fill of every activation-gradient buffer, at the start of the backward half"""
        self._add(self.lib.cp_memset_zero, lambda P: (self.grad_arena.data_ptr(), self.grad_arena.numel()), "grad_zero", [], [])

    # ---- tape
    def mark_forward_end(self):
        self.n_fwd_ops = len(self.ops)
        self.half = 1
        for kind in ("views", "packs"):
            self._add(self.lib.cp_pack_batch, (lambda k: lambda P: self._batch_args(k, 1))(kind), "prep_%s_bwd" % kind, [], [])

    def unwind(self):
        """append the backward launches; remember, per parameter, the op index after which its gradient is final"""
        for fn in reversed(self.tape):
            self._touched = []
            fn()
            for key in self._touched:
                self.pslot_done[key] = len(self.ops)
            self._wg_keys.update(self._touched)
        self.flush_wgrad()
        self.tape = []

    def gradient_buckets(self, nbuckets=4):
        """Cut the flat gradient buffer into ~equal contiguous buckets and the backward launches into as many segments, such
        that bucket k is final once segment k has run: [(call_lo, call_hi, elem_lo, elem_hi)], segments in launch order.  The
        backward runs head first, so the buffer's tail (decoder / refinement parameters) completes long before the
        backbone's -- its all-reduce can travel over xGMI under the remaining backward."""
        from .parallel import plan_gradient_buckets
        return plan_gradient_buckets(self.pslots, self.pgrad.numel(), self.pslot_done_call, self.n_fwd_ops, len(self.calls), nbuckets)

    def read_act(self, a: Act):
        """debug: the current contents of an activation / gradient view as a (B, H, W, C) fp32 CPU tensor"""
        dt = torch.bfloat16 if self.dtype == CP_BF16 else torch.float32
        es = 2 if self.dtype == CP_BF16 else 4
        n = a.B * a.H * a.W * a.cstride
        raw = a.tbuf.fixed if a.tbuf.fixed is not None else self.workspace[a.tbuf.offset:a.tbuf.offset + a.tbuf.nbytes]
        flat = raw[:n * es].view(dt).view(a.B, a.H, a.W, a.cstride)
        return flat[..., a.coff:a.coff + a.C].float().cpu()

    def run_lanes_range(self, streams, lo, hi):
        """launches [lo, hi) with the fork/join structure of the schedule (streams[0] being captured into a hipGraph)"""
        ptr = [st.cuda_stream for st in streams]
        active, events = 1, []

        def new_event():
            events.append(torch.cuda.Event())
            return events[-1]

        ci = 0
        for item in self.sched:
            kind = item[0]
            if kind == "op":
                if lo <= ci < hi:
                    fn, args, name = self.calls[item[1]]
                    rc = fn(ptr[item[2] if item[2] < len(ptr) else 0], *args[1:])
                    if rc != 0:
                        _abi.check(rc, name)
                ci += 1
                continue
            if not (lo <= ci < hi or (kind == "join" and ci == hi)):     # the join that closes the range's last region
                continue
            if kind == "fork":
                active = min(item[1], len(streams))
                ev = new_event()
                ev.record(streams[0])
                for k in range(1, active):
                    streams[k].wait_event(ev)
            elif kind == "sync":
                if item[1] < len(streams) and item[2] < len(streams):
                    ev = new_event()
                    ev.record(streams[item[1]])
                    streams[item[2]].wait_event(ev)
            elif kind in ("mark", "wait"):
                raise RuntimeError("mark/wait lane edges are not emitted in training programs")
            else:
                for k in range(1, active):
                    ev = new_event()
                    ev.record(streams[k])
                    streams[0].wait_event(ev)
                active = 1
        return events

    def run_prep_range(self, main, side, lo, hi, chunk=48):
        """launches [lo, hi) with the weight-preparation ops on `side`, chunks ahead of the compute ops on `main` (both
        torch.cuda.Stream; `main` is being captured).  Compute op j waits (one event per chunk) for every prep op < j."""
        prep = [i for i in range(lo, hi) if i in self.prep_calls]
        events = []
        ev0 = torch.cuda.Event()
        ev0.record(main)
        side.wait_event(ev0)
        events.append(ev0)
        launched, waited = 0, 0               # prep ops launched on `side` / covered by a wait on `main`
        chunk_ev = []                         # (number of prep ops covered, event)

        def launch_chunk():
            nonlocal launched
            end = min(launched + chunk, len(prep))
            for k in range(launched, end):
                fn, args, name = self.calls[prep[k]]
                rc = fn(side.cuda_stream, *args[1:])
                if rc != 0:
                    _abi.check(rc, name)
            launched = end
            ev = torch.cuda.Event()
            ev.record(side)
            chunk_ev.append((end, ev))
            events.append(ev)

        need = 0
        for i in range(lo, hi):
            if i in self.prep_calls:
                need += 1
                continue
            while launched < min(len(prep), need + chunk):      # keep the side stream at least one chunk ahead
                launch_chunk()
            if waited < need:
                for covered, ev in chunk_ev:
                    if covered >= need:
                        main.wait_event(ev)
                        waited = covered
                        break
            fn, args, name = self.calls[i]
            rc = fn(main.cuda_stream, *args[1:])
            if rc != 0:
                _abi.check(rc, name)
        while launched < len(prep):
            launch_chunk()
        if chunk_ev:                          # join: the capture must end with every forked stream merged back
            main.wait_event(chunk_ev[-1][1])
        return events

    def run_range(self, stream_ptr, lo, hi):
        for fn, args, name in self.calls[lo:hi]:
            rc = fn(stream_ptr, *args[1:])
            if rc != 0:
                _abi.check(rc, name)
