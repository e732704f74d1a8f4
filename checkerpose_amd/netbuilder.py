"""Emit the CheckerPose forward as a Program (engine.py) -- the host-side restatement of the op ORDER of
InitNet_GNN.forward (reference checkerpose/model/init.py:109-128) and PoseNet_GNNskip.forward
(model/pipeline.py:351-384; LM twin model/pipeline_lm.py:392-425).  No arithmetic happens here: every step
appends a C-ABI launch.  Layout: channels-last activations, graph features (B, N, C).
"""
import os

import torch

from ._abi import ACT_LEAKY, ACT_NONE, ACT_RELU
from .engine import Act, Program, _rup

from .model.backbone import HRNET_CFGS     # layer1 = (blocks, planes), stages = ((modules, blocks per branch, branch channels), ...)
IMG_FEATS_DIMS = {"resnet34": [64, 128, 256, 512], "hrnet_w18": [128, 256, 512, 1024], "hrnet_w18_small": [128, 256, 512, 1024],
                  "hrnet_w30": [128, 256, 512, 1024]}   # pipeline.py:6-15


FUSE_OUT_DEEP = os.environ.get("CHECKERPOSE_AMD_FUSE_OUT_DEEP", "1") != "0"   # A/B: 2nd / 3rd convs of the stride-2 fuse chains through cp_hr_fuse_out


DECODER_SPLITK = os.environ.get("CHECKERPOSE_AMD_DECODER_SPLITK", "1") != "0"   # A/B: decoder convs as split-K convs at batch 1-4
HEAD_ON_REFINE_LANE = os.environ.get("CHECKERPOSE_AMD_HEAD_LANE", "1") != "0"   # A/B: InitNet head beside up_net[0]
# opt-in (measured +-0: 11.02 / 11.02 vs 11.09 / 10.73 ms at B = 256, alternating on one box): the 2nd / 3rd convs of the long stride-2
# fuse chains run in the NEXT module's region, on their target's lane, instead of behind the 64 x 64 branch's lane in front of the join
FUSE_DEFER = os.environ.get("CHECKERPOSE_AMD_FUSE_DEFER", "0") == "1"
FUSE_SAME_LANE = os.environ.get("CHECKERPOSE_AMD_FUSE_SAME_LANE", "1") != "0"   # A/B: a fuse chain's later convs on its source's lane (no second region)


class _PendingTerm:
    """A fuse term whose stride-2 chain is only started: `act` = the output of its first conv, `convs` = [(conv key, bn key, relu)]
    still to run.  The next module resolves it on the lane of the branch that consumes it (NetEmitter._resolve)."""
    __slots__ = ("act", "convs")

    def __init__(self, act, convs):
        self.act, self.convs = act, convs


class NetEmitter:
    def __init__(self, prog: Program, sd):
        self.p, self.ws, self.sd = prog, prog.ws, sd
        self.ones = {}
        self.tp = prog if getattr(prog, "training", False) else None    # trainer.TrainProgram: emit forward + tape
        self.bn_counters = []                                            # num_batches_tracked buffers touched

    def W(self, key):
        return self.ws._w(key)

    def _unit(self, n):
        if n not in self.ones:
            self.ones[n] = torch.ones(n, dtype=torch.float32, device=self.p.device)
        return self.ones[n]

    # ---- conv + folded BN (+residual) (+ReLU)
    def conv_bn(self, x, conv, bn, k, stride, pad, relu=True, residual=None, out=None):
        w = self.W(conv + ".weight")
        if self.tp is not None:
            return self._conv_bn_train(x, conv, bn, w, k, stride, pad, ACT_RELU if relu else ACT_NONE, residual, out)
        s, t = self.ws.bn_fold(bn)
        return self.p.conv(x, conv, w, s, t, k, k, stride, pad, w.shape[0], ACT_RELU if relu else ACT_NONE,
                           residual=residual, out=out)

    # ---- train mode: raw conv -> batch statistics -> affine+act, backward pushed on the tape
    def _bn_train(self, raw, bn, act, residual, out, slope=0.0):
        tp = self.tp
        Cout = raw.C
        st = tp.bn_stats(raw, Cout, self.W(bn + ".weight"), self.W(bn + ".bias"), self.sd[bn + ".running_mean"],
                         self.sd[bn + ".running_var"])
        self.bn_counters.append(self.sd[bn + ".num_batches_tracked"])
        y = out if out is not None else tp.act(raw.H, raw.W, Cout)
        tp.bn_apply(raw, st, residual, y, act, slope)
        if act != ACT_NONE:
            tp.kinks[bn] = y
        return y, st

    def _conv_bn_train(self, x, conv, bn, w, k, stride, pad, act, residual, out):
        tp = self.tp
        Cout = w.shape[0]
        raw = tp.conv(x, conv, w, tp.const_vec(Cout, True), tp.const_vec(Cout, False), k, k, stride, pad, Cout)
        y, st = self._bn_train(raw, bn, act, residual, out)

        def bwd():
            if y.tbuf not in tp.grads:
                return
            gy = tp.grad_of(y)
            gres = tp.grad_of(residual) if residual is not None else None
            tp.bn_bwd(gy, y, raw, st, act, 0.0, gres, tp.pg_ptr(bn + ".weight"), tp.pg_ptr(bn + ".bias"))
            tp.conv_backward(conv, w, x, gy, k, k, stride, pad)
        tp.tape.append(bwd)
        return y

    def conv_bn_group(self, specs):
        """Training program: conv + train-mode BatchNorm (+residual) (+ReLU) of INDEPENDENT layers -- the convs one launch each, their
        BatchNorm passes grouped (trainer.bn_train_group / bn_bwd_group: one launch per pass for the whole group).
        specs: [(x, conv key, bn key, k, stride, pad, relu, residual)] -> [y]"""
        tp = self.tp
        members, recs = [], []
        ws = [self.W(conv + ".weight") for _, conv, *_ in specs]
        if all((k, stride, pad) == (3, 1, 1) for _, _, _, k, stride, pad, _, _ in specs):        # the convs themselves grouped too
            raws = tp.conv3x3_group([(x, conv, w, tp.const_vec(w.shape[0], True), tp.const_vec(w.shape[0], False), ACT_NONE, 0.0, None, None)
                                     for (x, conv, *_), w in zip(specs, ws)])
        else:
            raws = [tp.conv(x, conv, w, tp.const_vec(w.shape[0], True), tp.const_vec(w.shape[0], False), k, k, stride, pad, w.shape[0])
                    for (x, conv, _, k, stride, pad, _, _), w in zip(specs, ws)]
        for (x, conv, bn, k, stride, pad, relu, residual), w, raw in zip(specs, ws, raws):
            recs.append((x, conv, bn, w, k, stride, pad, ACT_RELU if relu else ACT_NONE, residual, raw))
        ys = []
        for x, conv, bn, w, k, stride, pad, act, residual, raw in recs:
            y = tp.act(raw.H, raw.W, raw.C)
            members.append((raw, raw.C, self.W(bn + ".weight"), self.W(bn + ".bias"), self.sd[bn + ".running_mean"],
                            self.sd[bn + ".running_var"], residual, y, act, 0.0))
            self.bn_counters.append(self.sd[bn + ".num_batches_tracked"])
            if act != ACT_NONE:
                tp.kinks[bn] = y
            ys.append(y)
        sts = tp.bn_train_group(members)

        def bwd():
            live = [(r, y, st) for r, y, st in zip(recs, ys, sts) if y.tbuf in tp.grads]
            if not live:
                return
            gys = [tp.grad_of(y) for _, y, _ in live]
            tp.bn_bwd_group([(gy, y, r[9], st, r[7], 0.0, tp.grad_of(r[8]) if r[8] is not None else None,
                              tp.pg_ptr(r[2] + ".weight"), tp.pg_ptr(r[2] + ".bias")) for gy, (r, y, st) in zip(gys, live)])
            tp.conv_backward_group([(r[1], r[3], r[0], gy, r[4], r[4], r[5], r[6]) for gy, (r, y, st) in zip(gys, live)])
        tp.tape.append(bwd)
        return ys

    def _hr_module_train(self, pfx, xs, nblocks=4):
        """hr_module for the training program, emitted DEPTH-major: block k / conv c of every branch, then ONE grouped launch per
        BatchNorm pass for all of them (the branches are independent until the fuse layers); likewise the fuse convs level by level"""
        nb = len(xs)
        xs = [self._materialize(x) for x in xs]
        for k in range(nblocks):
            bp = ["%s.branches.%d.%d" % (pfx, j, k) for j in range(nb)]
            assert not any((q + ".downsample.0.weight") in self.sd for q in bp)
            ys = self.conv_bn_group([(xs[j], bp[j] + ".conv1", bp[j] + ".bn1", 3, 1, 1, True, None) for j in range(nb)])
            xs = self.conv_bn_group([(ys[j], bp[j] + ".conv2", bp[j] + ".bn2", 3, 1, 1, True, xs[j]) for j in range(nb)])
        terms = [[None] * nb for _ in range(nb)]
        for i in range(nb):
            terms[i][i] = xs[i]
        cur = {}
        pairs = [(i, j) for j in range(nb) for i in range(nb) if i != j]
        level = 0
        while True:
            specs, who = [], []
            for i, j in pairs:
                q = "%s.fuse_layers.%d.%d" % (pfx, i, j)
                if j > i:
                    if level == 0:
                        specs.append((xs[j], q + ".0", q + ".1", 1, 1, 0, False, None))
                        who.append((i, j))
                elif level < i - j:
                    src = xs[j] if level == 0 else cur[(i, j)]
                    specs.append((src, "%s.%d.0" % (q, level), "%s.%d.1" % (q, level), 3, 2, 1, level != i - j - 1, None))
                    who.append((i, j))
            if not specs:
                break
            for (i, j), y in zip(who, self.conv_bn_group(specs)):
                cur[(i, j)] = y
            level += 1
        for i, j in pairs:
            terms[i][j] = cur[(i, j)]
        outs, recs = [], []
        for i in range(nb):
            out = self.p.act(xs[i].H, xs[i].W, xs[i].C)
            shifts = [max(j - i, 0) for j in range(nb)]
            outs.append(self.p.fuse_sum(terms[i], shifts, out, relu=True))
            self.tp.kinks["%s.fuse%d" % (pfx, i)] = out
            recs.append((out, list(terms[i]), shifts))
        tp = self.tp

        def bwd():                        # the whole fuse layer's backward in one launch: every (output, term) pair has its own gradient tensor
            mem = [(tp.grad_of(out), out, tp.grad_of(s_), sh, True) for out, srcs, shifts in recs if out.tbuf in tp.grads
                   for s_, sh in zip(srcs, shifts)]
            if mem:
                tp.fuse_sum_bwd_group(mem)
        tp.tape.append(bwd)
        return outs

    def _bias_vec(self, key, n):
        return self.tp.live_vec(n, [(0, self.W(key + ".bias"), 0, n)])

    def _linear_train(self, x, key, act, slope, out):
        tp = self.tp
        w = self.W(key + ".weight")
        Cout = w.shape[0]
        w4 = w.view(Cout, w.shape[1], 1, 1)
        y = tp.conv(x, key, w4, tp.const_vec(Cout, True), self._bias_vec(key, Cout), 1, 1, 1, 0, Cout, act, slope, out=out)
        if act != ACT_NONE:
            tp.kinks[key] = y

        def bwd():
            if y.tbuf not in tp.grads:
                return
            gy = tp.grad_of(y)
            tp.bn_bwd(gy, y, None, None, act, slope, None, None, tp.pg_ptr(key + ".bias"))
            tp.conv_backward(key, w4, x, gy, 1, 1, 1, 0)
        tp.tape.append(bwd)
        return y

    def _out_layer_bwd(self, key, w4, x, C_, src, base, sb, sp, sc, k=1, pad=0):
        """backward of a bias layer whose output went straight into a strided fp32 output block (logits / seg): gather the
        incoming gradient `src` (fp32 torch tensor, filled by the autograd hook) into channels-last, then bias/weight/data
        gradients as usual."""
        tp = self.tp

        def bwd():
            gy = tp.act(x.H + 2 * pad - k + 1, x.W + 2 * pad - k + 1, C_)
            tp.keep.append(src)
            tp.strided_to_act(lambda P: src.data_ptr(), 0, base, sb, sp, sc, gy, C_)
            tp.bn_bwd(gy, None, None, None, ACT_NONE, 0.0, None, None, tp.pg_ptr(key + ".bias"))
            tp.conv_backward(key, w4, x, gy, k, k, 1, pad)
        tp.tape.append(bwd)

    def linear(self, x, key, act=ACT_NONE, slope=0.0, out=None, **kw):
        """nn.Linear / 1x1 conv with bias over a (B, 1, N, C) view."""
        if self.tp is not None and not kw:
            return self._linear_train(x, key, act, slope, out)
        w = self.W(key + ".weight")
        w4 = w.reshape(w.shape[0], w.shape[1], 1, 1)
        b = self.W(key + ".bias")
        return self.p.conv(x, key, w4, self._unit(w.shape[0]), b, 1, 1, 1, 0, w.shape[0], act, slope, out=out, **kw)

    # ---- timm resnet blocks
    def basic_block(self, pfx, x, stride=1):
        w1 = self.W(pfx + ".conv1.weight")
        if (self.tp is None and stride == 1 and (pfx + ".downsample.0.weight") not in self.sd and w1.shape[0] == w1.shape[1]
                and self.p.can_fuse_basicblock(x, w1.shape[0])):
            s1, t1 = self.ws.bn_fold(pfx + ".bn1")
            s2, t2 = self.ws.bn_fold(pfx + ".bn2")
            return self.p.basicblock_fused(x, pfx + ".conv1", w1, s1, t1, pfx + ".conv2", self.W(pfx + ".conv2.weight"), s2, t2)
        y = self.conv_bn(x, pfx + ".conv1", pfx + ".bn1", 3, stride, 1)
        sc = x
        if (pfx + ".downsample.0.weight") in self.sd:
            sc = self.conv_bn(x, pfx + ".downsample.0", pfx + ".downsample.1", 1, stride, 0, relu=False)
        return self.conv_bn(y, pfx + ".conv2", pfx + ".bn2", 3, 1, 1, relu=True, residual=sc)

    def bottleneck(self, pfx, x, out=None):
        w1, w3 = self.W(pfx + ".conv1.weight"), self.W(pfx + ".conv3.weight")
        has_ds = (pfx + ".downsample.0.weight") in self.sd
        if (self.tp is None and self.p.can_fuse_bottleneck(x, w1.shape[0], w3.shape[0], has_ds) and w1.shape[1] == x.C
                and (has_ds or w1.shape[1] == w3.shape[0])):
            keys = [pfx + ".conv%d" % i for i in (1, 2, 3)]
            affs = [self.ws.bn_fold(pfx + ".bn%d" % i) for i in (1, 2, 3)]
            ws = [w1, self.W(pfx + ".conv2.weight"), w3]
            if has_ds:
                wd = self.W(pfx + ".downsample.0.weight")
                if tuple(wd.shape[2:]) != (1, 1) or wd.shape[0] != 256:
                    has_ds = None
                else:
                    keys.append(pfx + ".downsample.0")
                    affs.append(self.ws.bn_fold(pfx + ".downsample.1"))
                    ws.append(wd)
            if has_ds is not None:
                return self.p.bottleneck_fused(x, keys, ws, affs, out=out)
        y = self.conv_bn(x, pfx + ".conv1", pfx + ".bn1", 1, 1, 0)
        y = self.conv_bn(y, pfx + ".conv2", pfx + ".bn2", 3, 1, 1)
        sc = x
        if (pfx + ".downsample.0.weight") in self.sd:
            sc = self.conv_bn(x, pfx + ".downsample.0", pfx + ".downsample.1", 1, 1, 0, relu=False, out=None)
        return self.conv_bn(y, pfx + ".conv3", pfx + ".bn3", 1, 1, 0, relu=True, residual=sc, out=out)

    # ---- HRNet-W18 features (timm HighResolutionNetFeatures)
    def _resolve(self, t):
        """run what is left of a deferred fuse chain (on the CURRENT lane) -> its Act"""
        if not isinstance(t, _PendingTerm):
            return t
        a = t.act
        for ck, bk, relu in t.convs:
            if FUSE_OUT_DEEP and self.p.can_fuse_out(a, [self.W(ck + ".weight").shape[0]]):
                a = self.p.hr_fuse_out(a, [(ck, self.W(ck + ".weight")) + tuple(self.ws.bn_fold(bk)) + (3, relu)])[0]
            else:
                a = self.conv_bn(a, ck, bk, 3, 2, 1, relu=relu)
        return a

    def _materialize(self, x, out=None):
        """a module output that is still the un-summed list of fuse terms -> one activation (cp_fuse_sum_act), optionally
        written into the channel slice `out` of a wider buffer"""
        if isinstance(x, Act):
            assert out is None
            return x
        terms, shifts = x
        terms = [self._resolve(t) for t in terms]
        if out is None:
            out = self.p.act(terms[0].H << shifts[0], terms[0].W << shifts[0], terms[0].C)
        return self.p.fuse_sum(terms, shifts, out, relu=True)

    def bottleneck_incre(self, pfx, cat: Act, planes, out=None):
        """timm Bottleneck with a projection shortcut (HRNet incre_modules) whose input x already sits in channels
        [planes, planes + C) of `cat`: conv2's output goes into channels [0, planes) of the same buffer and the block's tail
            relu(bn3(conv3(t2)) + bn_d(conv_d(x)))  =  relu([s3 W3 | sd Wd] [t2 ; x] + (t3 + td))
        is ONE 1x1 conv over the concatenated channels -- the shortcut conv, its 4*planes-channel output tensor and the
        residual read disappear (the 64x64 one alone moved 0.8 GB per step at batch 256)."""
        x = cat.slice(planes, cat.C - planes, _rup(cat.C - planes, self.p.E))
        y = self.conv_bn(x, pfx + ".conv1", pfx + ".bn1", 1, 1, 0)
        self.conv_bn(y, pfx + ".conv2", pfx + ".bn2", 3, 1, 1, out=cat.slice(0, planes))
        ck = pfx + ".conv3#merged"
        if ck not in self.ws.cache:
            s3, t3 = self.ws.bn_fold(pfx + ".bn3")
            sd, td = self.ws.bn_fold(pfx + ".downsample.1")
            w3, wd = self.W(pfx + ".conv3.weight"), self.W(pfx + ".downsample.0.weight")
            self.ws.cache[ck] = (torch.cat([w3 * s3.view(-1, 1, 1, 1), wd * sd.view(-1, 1, 1, 1)], dim=1).contiguous(), t3 + td)
        wm, shift = self.ws.cache[ck]
        return self.p.conv(cat, ck, wm, self._unit(wm.shape[0]), shift, 1, 1, 1, 0, wm.shape[0], ACT_RELU, out=out)

    def hr_module(self, pfx, xs, lazy=False, defer=False, nblocks=4):
        """One timm HighResolutionModule.  xs[j]: an Act, or (terms, shifts) = the previous module's fuse terms of branch j
        still un-summed.  With lazy=True the outputs are returned in that un-summed form too: a branch that runs as ONE
        chain launch (engine.hr_chain, bf16) sums + ReLUs them while staging its map into LDS, other consumers
        materialise them (fuse_sum launch)."""
        nb = len(xs)
        if self.tp is not None and self.tp.bn_grouped and nb > 1:
            return self._hr_module_train(pfx, xs, nblocks)
        xs = list(xs)
        p = self.p
        # lane j: branch j's four BasicBlocks; then lane i runs the fuse-layer conv chains INTO branch i (it waits for the
        # source branch j): the long stride-2 chains fed by the 64x64 branch -- the module's critical path -- then run beside
        # each other on the lanes whose own (low-resolution) branch finished early, not one after the other behind branch 0
        terms = [[None] * nb for _ in range(nb)]
        first = {}                           # (i, j) -> output of fuse chain j -> i's first conv, from the grouped launch
        p.par_begin(nb)
        for j in range(nb):
            p.set_lane(j)
            if not isinstance(xs[j], Act):       # deferred tails of the previous module's stride-2 fuse chains INTO branch j: on this lane,
                xs[j] = ([self._resolve(t) for t in xs[j][0]], xs[j][1])     # in front of its chain (branches 0 / 1 have none and start at once)
            if isinstance(xs[j], Act):
                C_, H, W = xs[j].C, xs[j].H, xs[j].W
            else:
                t0, sh0 = xs[j][0][0], xs[j][1][0]
                C_, H, W = t0.C, t0.H << sh0, t0.W << sh0
            if self.tp is None and nblocks == 4 and p.can_chain(C_, H, W):      # (the chain launches are 4 BasicBlocks = 8 convs long)
                srcs, shifts = ([xs[j]], [0]) if isinstance(xs[j], Act) else xs[j]
                bp = "%s.branches.%d" % (pfx, j)
                ws = [self.W("%s.%d.conv%d.weight" % (bp, k, c)) for k in range(4) for c in (1, 2)]
                affs = [self.ws.bn_fold("%s.%d.bn%d" % (bp, k, c)) for k in range(4) for c in (1, 2)]
                tl = [("%s.fuse_layers.%d.%d.0.0" % (pfx, i, j), "%s.fuse_layers.%d.%d.0.1" % (pfx, i, j), i - j > 1) for i in range(j + 1, nb)]
                tconvs = [(ck, self.W(ck + ".weight")) + tuple(self.ws.bn_fold(bk)) + (relu,) for ck, bk, relu in tl]
                if tl and all(i > j for i in range(nb) if i != j) and p.can_chain_tail(C_, H, W, tconvs):
                    # every first-level fuse conv fed by this branch is a stride-2 one (branch 0): the chain launch runs them off its
                    # finished map in LDS -- the grouped launch behind the module's longest chain and its re-read of the map go away
                    xs[j], touts = p.hr_chain(bp, srcs, shifts, not isinstance(xs[j], Act), ws, affs, C_, H, W, tail=tconvs)
                    for i, o in zip(range(j + 1, nb), touts):
                        first[(i, j)] = o
                    continue
                # (round 5) the other chains likewise: every first-level fuse conv that reads branch j -- the 1x1 convs towards the
                # higher-resolution branches and the first stride-2 conv towards each lower-resolution one -- in the chain launch's tail
                lst_t = []
                for i in range(nb):
                    q = "%s.fuse_layers.%d.%d" % (pfx, i, j)
                    if j > i:
                        lst_t.append((i, q + ".0", q + ".1", 1, False))
                    elif j < i:
                        lst_t.append((i, q + ".0.0", q + ".0.1", 3, i - j > 1))
                tcv = [(ck, self.W(ck + ".weight")) + tuple(self.ws.bn_fold(bk)) + (k, relu) for (_, ck, bk, k, relu) in lst_t]
                if nb > 1 and p.can_chain_tails(C_, H, W, tcv):
                    xs[j], touts = p.hr_chain_tails(bp, srcs, shifts, not isinstance(xs[j], Act), ws, affs, C_, H, W, tcv)
                    for (i, _, _, _, _), o in zip(lst_t, touts):
                        first[(i, j)] = o
                    continue
                xs[j] = p.hr_chain(bp, srcs, shifts, not isinstance(xs[j], Act), ws, affs, C_, H, W)
            else:
                xs[j] = self._materialize(xs[j])
                for k in range(nblocks):
                    xs[j] = self.basic_block("%s.branches.%d.%d" % (pfx, j, k), xs[j])
            lst = []                     # every first-level fuse conv that reads branch j
            for i in range(nb):
                q = "%s.fuse_layers.%d.%d" % (pfx, i, j)
                if j > i:
                    lst.append((i, q + ".0", q + ".1", 1, False))
                elif j < i:
                    lst.append((i, q + ".0.0", q + ".0.1", 3, i - j > 1))
            if self.tp is None and nb > 1 and p.can_fuse_out(xs[j], [self.W(ck + ".weight").shape[0] for (_, ck, _, _, _) in lst]):
                # the same lane goes on with ONE launch for all of them (its map staged in LDS once); the second region keeps only the
                # 2nd / 3rd convs of the long stride-2 chains
                outs = p.hr_fuse_out(xs[j], [(ck, self.W(ck + ".weight")) + tuple(self.ws.bn_fold(bk)) + (k, relu)
                                             for (_, ck, bk, k, relu) in lst])
                for (i, _, _, _, _), o in zip(lst, outs):
                    first[(i, j)] = o
        if self.tp is not None:              # training program: lane j also runs the fuse chains fed by branch j (program order)
            sched = [(j, i, j) for j in range(nb) for i in range(nb) if i != j]
        else:
            # eval: the fuse-layer conv chains run in a second fork/join region, spread over the lanes by length (the three
            # stride-2 chains fed by the 64x64 branch used to queue up behind it on ONE lane: the module's critical path).
            # (Cross-lane event edges inside one region would express this without the join, but a capture in which two
            # streams wait on each other's events crashes hipStreamEndCapture on ROCm 7.2.)
            def cost_of(i, j):               # launches still to run for term (i, j)
                if (i, j) in first:
                    return max(i - j - 1, 0)
                return (i - j) if j < i else 0.3
            load = [0.0] * nb
            sched = []
            same_lane = FUSE_SAME_LANE and len(first) == nb * (nb - 1)      # every chain starts in a grouped launch:
            for cost, i, j in sorted([(cost_of(i, j), i, j) for i in range(nb) for j in range(nb) if i != j], reverse=True):
                if same_lane:                # its later convs follow on their SOURCE's lane -- one fork/join region per module
                    ln = j
                else:
                    ln = min(range(nb), key=lambda k: load[k]) if cost > 0 else 0
                load[ln] += cost
                sched.append((ln, i, j))
            second = any(c > 0 for c in load) and not same_lane
            if second:
                p.par_end()
                p.par_begin(nb)
        for i in range(nb):
            terms[i][i] = xs[i]
        for ln, i, j in sched:
            p.set_lane(ln)
            q = "%s.fuse_layers.%d.%d" % (pfx, i, j)
            if (i, j) in first:
                # the rest of a long stride-2 chain (same kernel, one conv per launch): now, on its source's lane -- or, when the
                # consumer is the next module of the stage, deferred to that module's region and its TARGET's lane: the 64 x 64
                # branch's lane is the module's critical path (chain + the grouped first-level launch), and its three small tail convs
                # used to run behind it, in front of the join
                t = _PendingTerm(first[(i, j)], [("%s.%d.0" % (q, k), "%s.%d.1" % (q, k), k != i - j - 1) for k in range(1, i - j)])
                if not (self.tp is None and defer and lazy and same_lane and t.convs):
                    t = self._resolve(t)
            elif j > i:
                t = self.conv_bn(xs[j], q + ".0", q + ".1", 1, 1, 0, relu=False)
            else:
                t = xs[j]
                for k in range(i - j):
                    t = self.conv_bn(t, "%s.%d.0" % (q, k), "%s.%d.1" % (q, k), 3, 2, 1, relu=(k != i - j - 1))
            terms[i][j] = t
        p.par_end()
        if lazy and self.tp is None:
            return [(terms[i], [max(j - i, 0) for j in range(nb)]) for i in range(nb)]
        outs = []
        p.par_begin(nb)                      # the nb fuse sums are independent of each other
        for i in range(nb):
            p.set_lane(i)
            out = p.act(xs[i].H, xs[i].W, xs[i].C)
            shifts = [max(j - i, 0) for j in range(nb)]
            outs.append(p.fuse_sum(terms[i], shifts, out, relu=True))
            if self.tp is not None:
                self.tp.kinks["%s.fuse%d" % (pfx, i)] = out
                self._fuse_sum_tape(out, list(terms[i]), shifts)
        p.par_end()
        return outs

    def _fuse_sum_tape(self, out, srcs, shifts):
        tp = self.tp

        def bwd():
            if out.tbuf not in tp.grads:
                return
            go = tp.grad_of(out)
            for s_, sh in zip(srcs, shifts):
                tp.fuse_sum_bwd(go, out, tp.grad_of(s_), sh, True)
        tp.tape.append(bwd)

    def hrnet(self, pfx, x, feat_outs=None, stem_done=False, name="hrnet_w18"):
        hcfg = HRNET_CFGS[name]
        stages = [("stage%d" % (si + 2),) + tuple(st) for si, st in enumerate(hcfg["stages"])]      # (stage, modules, blocks per branch, channels)
        if not stem_done:
            x = self.conv_bn(x, pfx + "conv1", pfx + "bn1", 3, 2, 1)
            x = self.conv_bn(x, pfx + "conv2", pfx + "bn2", 3, 2, 1)
        for k in range(hcfg["layer1"][0]):
            x = self.bottleneck("%slayer1.%d" % (pfx, k), x)
        if self.tp is None:                  # the two transition convs are independent: two lanes
            self.p.par_begin(2)
            xs = [self.conv_bn(x, pfx + "transition1.0.0", pfx + "transition1.0.1", 3, 1, 1)]
            self.p.set_lane(1)
            xs.append(self.conv_bn(x, pfx + "transition1.1.0.0", pfx + "transition1.1.0.1", 3, 2, 1))
            self.p.par_end()
        else:
            xs = [self.conv_bn(x, pfx + "transition1.0.0", pfx + "transition1.0.1", 3, 1, 1),
                  self.conv_bn(x, pfx + "transition1.1.0.0", pfx + "transition1.1.0.1", 3, 2, 1)]
        for si, (stage, nmod, nblk, chans) in enumerate(stages):
            if si > 0:
                t = "%stransition%d.%d.0" % (pfx, si + 1, len(chans) - 1)
                xs = xs + [self.conv_bn(xs[-1], t + ".0", t + ".1", 3, 2, 1)]
            last_stage = si == len(stages) - 1
            for m in range(nmod):           # inside a stage the fuse sums stay un-summed for the next module's chain launches
                xs = self.hr_module("%s%s.%d" % (pfx, stage, m), xs, lazy=(m + 1 < nmod) or (last_stage and self.tp is None),
                                    defer=FUSE_DEFER and m + 1 < nmod, nblocks=nblk)
        feats = []
        self.p.par_begin(len(xs))            # the four incre bottlenecks are independent
        for i, f in enumerate(xs):
            self.p.set_lane(i)
            q = "%sincre_modules.%d.0" % (pfx, i)
            if isinstance(f, Act):
                feats.append(self.bottleneck(q, f, out=(feat_outs[i] if feat_outs else None)))
            else:                            # eval: the last module's fuse sum lands right beside the bottleneck's conv2 output
                terms, shifts = f
                planes = self.W(q + ".conv1.weight").shape[0]
                cat = self.p.act(terms[0].H << shifts[0], terms[0].W << shifts[0], planes + terms[0].C)
                self._materialize(f, out=cat.slice(planes, terms[0].C, _rup(terms[0].C, self.p.E)))
                feats.append(self.bottleneck_incre(q, cat, planes, out=(feat_outs[i] if feat_outs else None)))
        self.p.par_end()
        return feats

    def resnet34(self, pfx, x, feat_outs=None):
        x = self.conv_bn(x, pfx + "conv1", pfx + "bn1", 7, 2, 3)
        x0 = x
        x = self.p.maxpool(x0)
        if self.tp is not None:
            tp, xo = self.tp, x

            def bwd_pool():
                if xo.tbuf in tp.grads:
                    tp.maxpool_bwd(x0, tp.grad_of(xo), tp.grad_of(x0))
            tp.tape.append(bwd_pool)
        feats = []
        for li, nblk in enumerate((3, 4, 6, 3)):
            for k in range(nblk):
                x = self.basic_block("%slayer%d.%d" % (pfx, li + 1, k), x, stride=2 if (k == 0 and li > 0) else 1)
            feats.append(x)
        return feats

    # ---- EdgeConv: per-node GEMM to [P'|Q'] + neighbour gather-max (factored StaticGraph_module)
    def _edgeconv_train(self, pfx, x, graph, slope, out):
        tp = self.tp
        w = self.W(pfx + ".conv.0.weight")            # (C', 2C, 1, 1)
        Co, Cc = w.shape[0], w.shape[1] // 2
        bn = pfx + ".conv.1"
        wpq = tp.edge_weight_view(w, Co, Cc, 0)       # (2C', C, 1, 1) = [W1 ; W2 - W1], rebuilt from the live weight
        pq = tp.conv(x, pfx + ".conv.0#pq", wpq, tp.const_vec(2 * Co, True), tp.const_vec(2 * Co, False), 1, 1, 1, 0, 2 * Co)
        if out is None:
            out = tp.act(1, x.W, Co)
        st = tp.edge_train_fwd(pq, graph, self.W(bn + ".weight"), self.W(bn + ".bias"), self.sd[bn + ".running_mean"],
                               self.sd[bn + ".running_var"], out, Co, slope)
        self.bn_counters.append(self.sd[bn + ".num_batches_tracked"])
        tp.kinks[pfx] = out

        def bwd():
            if out.tbuf not in tp.grads:
                return
            D = tp.act(1, x.W, 2 * Co)                # [dP - dQ | dQ]
            tp.debug[pfx] = dict(x=x, pq=pq, out=out, D=D, gout=tp.grad_of(out), st=st)
            tp.edge_train_bwd(pq, graph, st, out, tp.grad_of(out), D, tp.pg_ptr(bn + ".weight"), tp.pg_ptr(bn + ".bias"), Co, slope)
            dw = tp.pg_ptr(pfx + ".conv.0.weight")    # (C', 2C): dW1 = D1^T x, dW2 = D2^T x
            tp.wgrad(D.slice(0, Co), x, dw, Co, Cc, 1, 1, 1, 0, base=0, sco=2 * Cc, sci=1)
            tp.wgrad(D.slice(Co, Co), x, dw, Co, Cc, 1, 1, 1, 0, base=Cc, sco=2 * Cc, sci=1)
            if tp.needs_grad(x):
                gx = tp.grad_of(x)
                wd = tp.edge_weight_view(w, Co, Cc, 1)    # (C, 2C', 1, 1)
                tp.conv(D, pfx + ".conv.0#dgrad", wd, tp.const_vec(Cc, True), tp.const_vec(Cc, False), 1, 1, 1, 0, Cc,
                        residual=gx, out=gx)
        tp.tape.append(bwd)
        return out

    def edgeconv(self, pfx, x: Act, graph, slope, out: Act = None):
        if self.tp is not None:
            return self._edgeconv_train(pfx, x, graph, slope, out)
        w = self.W(pfx + ".conv.0.weight")            # (C', 2C, 1, 1)
        Co, C2 = w.shape[0], w.shape[1]
        Cc = C2 // 2
        ck = pfx + ".conv.0#pq"
        if ck not in self.ws.cache:
            w1, w2 = w[:, :Cc, 0, 0], w[:, Cc:, 0, 0]
            s, t = self.ws.bn_fold(pfx + ".conv.1")
            self.ws.cache[ck] = (torch.cat([w1, w2 - w1], 0).reshape(2 * Co, Cc, 1, 1).contiguous(),
                                 torch.cat([s, s]), torch.cat([torch.zeros_like(t), t]))
        wpq, sc, sh = self.ws.cache[ck]
        if out is None:
            out = self.p.act(1, x.W, Co)
        if self.p.can_fuse_edgeconv(x.W, graph["K"], Cc, Co) and x.C == Cc:
            return self.p.edge_fused(x, ck, wpq, sc, sh, graph["idx"], graph["gids"], out, graph["K"], graph["G"], slope)
        tl = graph.get("tiled")
        if tl is not None and x.C == Cc and self.p.can_tile_edgeconv(x.W, graph["K"], Cc, Co, tl["HPAD"]):
            return self.p.edge_tiled(x, ck, wpq, sc, sh, tl, graph["gids"], out, graph["K"], graph["G"], slope)
        pq = self.p.conv(x, ck, wpq, sc, sh, 1, 1, 1, 0, 2 * Co, in_half=self.p.gnn_half)     # (half in -> half out: the whole layer in the group's type)
        return self.p.edge_gather(pq, graph["idx"], graph["gids"], out, graph["K"], Co, graph["G"], slope)


def emit_init_net(em: NetEmitter, cfg, io, pfx="", graph_out: Act = None, feat_outs=None, defer_head=False):
    """InitNet_GNN.forward init.py:109-128.  Returns (feats [Act], graph_feats Act).  defer_head: only the backbone is emitted
    now and the second return value is a callable that emits the head (conv1x1 -> EdgeConv layers -> Linear) and returns the
    graph features -- PoseNet runs it on the refinement lane, beside the decoder's first stage (which needs the backbone only)."""
    p = em.p
    N = cfg["npoint"]
    tp = em.tp
    bb = pfx + "img_backbone."
    if cfg.get("inject_feats"):
        # test hook (HipForwardMixin._run(inject_feats=...)): the backbone's four features are GIVEN as NCHW fp32 tensors --
        # what the reference-made `*_injected` goldens do through their timm stub -- so the head (everything the reference's
        # own code pins) is compared with the reference directly; a feature the decoder expects inside a concat buffer is
        # copied into its channel slice (a one-term, no-ReLU cp_fuse_sum_act)
        assert tp is None
        feats = []
        for i, t in enumerate(io["inject"]):
            a = p.nchw_to_nhwc(t, t.shape[1], t.shape[2], t.shape[3])
            if feat_outs and feat_outs[i] is not None:
                a = p.fuse_sum([a], [0], feat_outs[i], relu=False)
            feats.append(a)
        if defer_head:
            return feats, (lambda: _emit_init_head(em, cfg, io, pfx, graph_out, feats))
        return feats, _emit_init_head(em, cfg, io, pfx, graph_out, feats)
    fused_stem = (tp is None and cfg["backbone"] in HRNET_CFGS and not cfg.get("uint8_input") and p.can_fuse_stem(cfg["img_size"]))    # stem_width 64 in all of them
    if fused_stem:                 # layout change + conv1 + conv2 of the HRNet stem in one launch, straight from the NCHW image
        s1, t1 = em.ws.bn_fold(bb + "bn1")
        s2, t2 = em.ws.bn_fold(bb + "bn2")
        x = p.hr_stem(io["img"], cfg["img_size"], bb + "conv1", em.W(bb + "conv1.weight"), s1, t1, bb + "conv2", em.W(bb + "conv2.weight"), s2, t2)
    elif cfg.get("uint8_input"):   # raw uint8 HWC crops: ToTensor + Normalize on the device (row N3)
        x = p.u8_to_nhwc_norm(io["img"], cfg["img_size"], cfg["img_size"])
    else:
        x = p.nchw_to_nhwc(io["img"], 3, cfg["img_size"], cfg["img_size"])
    if tp is not None:
        tp.nograd.add(id(x.tbuf))                           # the image needs no gradient
    feats = em.hrnet(bb, x, feat_outs=feat_outs, stem_done=fused_stem, name=cfg["backbone"]) if cfg["backbone"] in HRNET_CFGS else em.resnet34(bb, x)
    if defer_head:
        return feats, (lambda: _emit_init_head(em, cfg, io, pfx, graph_out, feats))
    return feats, _emit_init_head(em, cfg, io, pfx, graph_out, feats)


def _emit_init_head(em: NetEmitter, cfg, io, pfx, graph_out, feats):
    p, tp, N = em.p, em.tp, cfg["npoint"]
    f = feats[-1]                                           # (B, 8, 8, Cb)
    # conv1x1 Cb -> N, then `view(-1, N, 64).permute(0,2,1)` (init.py:112-114): keypoint n's 8x8 response map is its
    # 64-d feature -> written straight into the (B, N, 64) graph layout through the epilogue strides.
    nc1 = cfg.get("num_conv1x1", 1)
    c1key = pfx + ("conv1x1" if nc1 == 1 else "conv1x1.%d" % (2 * (nc1 - 1)))
    for j in range(nc1 - 1):     # init.py:87-95: Conv2d -> [LeakyReLU(0.01) -> Conv2d(N -> N)]*: the activation rides in the producer
        f = em.linear(f, pfx + "conv1x1.%d" % (2 * j), ACT_LEAKY, 0.01)
    w = em.W(c1key + ".weight")
    ng = cfg["init_num_graph_module"]
    tiled = tp is None and io["graph"].get("tiled") is not None
    # init_network_num_graph_module = 0 (config/lm/*woEdgeConv*, init_gnn0_*): the reinterpreted conv1x1 rows ARE the graph feature
    # (init.py:112-118 with an empty pre_query_block) -- inside PoseNet they go straight into their slice of stage 0's input rows
    direct = ng == 0 and graph_out is not None and not tiled
    g0 = graph_out if direct else p.act(1, N, 64)
    npix = f.H * f.W
    assert npix == g0.C, "conv1x1's response map is the keypoint's feature vector (init.py:114)"
    ostr = (g0.coff, N * g0.cstride, f.W, 1, g0.cstride)      # element (b, pixel, n) -> row n of crop b, channel = pixel
    if tp is None:      # (p.gnn_half: from here to the logits the rows are IEEE half -- engine.USE_GNN_F16)
        p.conv(f, c1key, w, em._unit(N), em.W(c1key + ".bias"), 1, 1, 1, 0, N,
               ostr=ostr, out_tbuf=g0.tbuf, out_half=p.gnn_half)
    else:
        p.conv(f, c1key, w, tp.const_vec(N, True), em._bias_vec(c1key, N), 1, 1, 1, 0, N,
               ostr=ostr, out_tbuf=g0.tbuf)
        f_in = f

        def bwd_conv1x1():
            if g0.tbuf not in tp.grads:
                return
            gg = tp.grad_of(g0)                              # (B, N, 64): gy[b, pix, n] = gg[b, n, pix]
            gy = tp.act(f_in.H, f_in.W, N)
            ggt = gg.tbuf
            tp.strided_to_act(lambda P: P(ggt), tp.dtype, gg.coff, N * gg.cstride, 1, gg.cstride, gy, N, reads=[ggt])
            tp.bn_bwd(gy, None, None, None, ACT_NONE, 0.0, None, None, tp.pg_ptr(c1key + ".bias"))
            tp.conv_backward(c1key, w, f_in, gy, 1, 1, 1, 0)
        tp.tape.append(bwd_conv1x1)
    g = g0
    if tiled:
        # large graphs (N > 512): from here on the rows follow the INTERNAL keypoint numbering in which 512 consecutive rows are a
        # compact patch of the kNN graph (graph_sched.tile_schedule); every op between here and the logits is per keypoint or
        # goes through the (renumbered) graph, and the runtime un-permutes the logit block / ids behind the last launch
        g = p.permute_rows(g0, p.act(1, N, 64), io["graph"]["tiled"]["perm"], io["graph"]["gids"])
    for i in range(ng):
        last = i == ng - 1
        g = em.edgeconv("%spre_query_block.%d" % (pfx, i), g, io["graph"], cfg["init_graph_slope"],
                        out=graph_out if (last and graph_out is not None) else None)
    if ng == 0 and graph_out is not None and not direct:
        raise RuntimeError("init_network_num_graph_module == 0 with more than 512 keypoints (patch-ordered rows) is not supported")
    # Linear(64 -> 1 + 2r) (init.py:107,120-122) into the (B,13,N) logit block: r = 3 -> rows [roi | x2 x1 x0 | . . . | y2 y1 y0]
    # (the layout PoseNet appends its refinement bits to); an InitNet used alone with another res_log2 writes rows [0, 1 + 2r)
    wl = em.W(pfx + "mlp.weight")
    nbits = wl.shape[0]
    r3 = nbits == 7
    rows = 10 if r3 else nbits
    rmap = [0, 1, 2, 3, -1, -1, -1, 4, 5, 6] if r3 else None
    bl = em.W(pfx + "mlp.bias")
    if tp is None:
        scv = em._unit(rows)
        shv = torch.cat([bl[:4], torch.zeros(3, device=p.device), bl[4:]]) if r3 else bl
    else:
        scv = tp.const_vec(rows, True)
        shv = tp.live_vec(10, [(0, bl, 0, 4), (7, bl, 4, 3)]) if r3 else tp.live_vec(rows, [(0, bl, 0, rows)])
    p.conv(g, pfx + "mlp", wl.view(nbits, 64, 1, 1), scv, shv,
           1, 1, 1, 0, rows, row_map=rmap, cout_rows=rows, out_f32=True,
           ostr=(0, 13 * N, 0, 1, N), out_tbuf=io["bits_tb"], in_half=(tp is None and p.gnn_half))
    if tp is not None:       # incoming gradient: rows [roi, x.., y..] gathered by the autograd hook into (B, 1 + 2r, N)
        em._out_layer_bwd(pfx + "mlp", wl.view(nbits, 64, 1, 1), g, nbits, io["dinit"], 0, nbits * N, 1, N)
    return g


def _convt_train_tail(em, up, wt, f, o, nf):
    """train mode: BatchNorm (batch statistics) + ReLU behind the 4 raw ConvTranspose2d phase launches, and the layer's
    backward: weight gradient = conv wgrad with the layer input as `dy` (coarse grid) and the output gradient as `x`;
    data gradient = plain 3x3/s2 conv over the output gradient with w read as (Cout' = Cin_t, Cin' = Cout_t, 3, 3)."""
    tp = em.tp
    y, st = em._bn_train(o, up + ".1", ACT_RELU, None, None)

    def bwd():
        if y.tbuf not in tp.grads:
            return
        gy = tp.grad_of(y)
        tp.bn_bwd(gy, y, o, st, ACT_RELU, 0.0, None, tp.pg_ptr(up + ".1.weight"), tp.pg_ptr(up + ".1.bias"))
        tp.wgrad(f, gy, tp.pg_ptr(up + ".0.weight"), wt.shape[0], wt.shape[1], 3, 3, 2, 1)
        gf = tp.grad_of(f)
        tp.conv(gy, up + ".0#dgrad", wt, tp.const_vec(wt.shape[0], True), tp.const_vec(wt.shape[0], False), 3, 3, 2, 1,
                wt.shape[0], residual=gf, out=gf)
    tp.tape.append(bwd)
    return y


def _upsample_tape(tp, src, dst_slice):
    def bwd():
        if dst_slice.tbuf not in tp.grads:
            return
        tp.upsample2x_bwd(tp.grad_of(dst_slice), tp.grad_of(src))
    tp.tape.append(bwd)


def _patch_tape(em, pgk, wpg, f, patches, lslice, io, N, Ech, k):
    """backward of Index2Feat (scatter-add of the keypoint gradients into an fp32 patch-map gradient, then into the
    storage dtype) followed by the patch_generator conv's bias / weight / data gradients."""
    tp = em.tp
    dpatch = torch.empty(patches.B * patches.H * patches.W * Ech, dtype=torch.float32, device=tp.device)
    tp.keep += [dpatch, io["xid"], io["yid"]]

    def bwd():
        if lslice.tbuf not in tp.grads:
            return
        tp.index2feat_bwd(tp.grad_of(lslice), io["xid"], io["yid"], io["mask"], dpatch, N, patches.H, patches.W, Ech, k)
        gp = tp.act(patches.H, patches.W, Ech)
        tp.strided_to_act(lambda P: dpatch.data_ptr(), 0, 0, patches.H * patches.W * Ech, Ech, 1, gp, Ech)
        tp.bn_bwd(gp, None, None, None, ACT_NONE, 0.0, None, None, tp.pg_ptr(pgk + ".bias"))
        tp.conv_backward(pgk, wpg, f, gp, k, k, 1, k - 1)
    tp.tape.append(bwd)


def emit_posenet(em: NetEmitter, cfg, io):
    """PoseNet_GNNskip.forward pipeline.py:351-384."""
    p = em.p
    N, B = cfg["npoint"], p.B
    nref = cfg["res_log2"] - 3
    active = cfg["stage"] if cfg.get("stage") is not None else nref
    ngs = cfg["num_graph_module"]
    ngs = (ngs,) * nref if isinstance(ngs, int) else tuple(ngs)
    nf = cfg["num_filters"]
    qd = cfg["query_dims"] or (nf, 256, 64)
    k = cfg["local_k"]
    slope = cfg["leaky_slope"]

    tp = em.tp

    def local_buf(i):
        gdim = 64 if i == 0 else qd[0]
        return p.act(1, N, qd[0] + gdim)       # [local 4*E | previous graph feature]

    L = local_buf(0) if active > 0 else None
    # eval: up_net[i >= 1] = conv(bilinear_x2(cat[img_feat, skip])) runs as ONE launch that interpolates inside its halo loader
    # (engine.conv_up2x); its input is the LOW-resolution concat buffer, filled in place by its two producers -- the previous
    # stage's last conv (channels [0, nf)) and the backbone's incre module (the skip feature, channels [nf, ..))
    lowcats, feat_outs = {}, None
    if tp is None and cfg["backbone"] in HRNET_CFGS:
        feat_outs = [None] * 4
        for i in range(1, active):
            j = 3 - i
            Cs = em.W("init_net.img_backbone.incre_modules.%d.0.conv3.weight" % j).shape[0]
            Hs = cfg["img_size"] // (4 << j)
            # (at batch 1-2 the split-K conv on a materialised upsample beats the fused loader's 8-32 workgroups)
            if p.can_conv_up2x(Hs, Hs, nf) and not (DECODER_SPLITK and p.would_splitk(p.B * 4 * Hs * Hs, 9 * (_rup(nf, p.E) + _rup(Cs, p.E)), nf)):
                lowcats[i] = p.act(Hs, Hs, nf + Cs)
                feat_outs[j] = lowcats[i].slice(_rup(nf, p.E), Cs)
    head_later = tp is None and active > 0 and HEAD_ON_REFINE_LANE
    feats, g = emit_init_net(em, cfg, io, "init_net.", graph_out=L.slice(qd[0], 64) if L is not None else None,
                             feat_outs=feat_outs, defer_head=head_later)
    dbits = io.get("decode_bits", io["bits"])      # teacher forcing (tests only): decode from supplied logits
    if not head_later:
        p.decode(dbits, -1, io["mask"], io["xid"], io["yid"], io["x64"], io["y64"], N)
    f = feats[-1]
    # lane 0: decoder chain up_net[0..2] -> seg (MFMA-bound) ; lane 1: refine stages (latency-bound graph kernels).
    # refine[i] needs up_net[i]'s output (sync 0 -> 1) and refine[i-1]; up_net[i+1] needs only up_net[i].
    if active > 0:
        p.par_begin(6 if tp is None else 2)      # 0 decoder, 1 refinement, 2 skip upsamples, 3-5 transposed-conv phases
    if head_later:     # InitNet's head needs only the backbone's last feature, like up_net[0]: it runs on the refinement lane beside it
        p.set_lane(1)
        g = g()
        p.decode(dbits, -1, io["mask"], io["xid"], io["yid"], io["x64"], io["y64"], N)
        p.set_lane(0)
    # the skip features' bilinear x2 halves of the decoder's concat buffers do not depend on the decoder: they run on a lane
    # of their own right away (lane 2 never waits for anybody) instead of on the decoder's critical path
    cats = {}
    if tp is None:
        for i in range(1, active):
            if i in lowcats:
                continue
            sk = feats[-i - 1]
            cats[i] = p.act(2 * sk.H, 2 * sk.W, nf + sk.C)
            p.set_lane(2)
            p.upsample2x(sk, cats[i].slice(_rup(nf, p.E), sk.C))
    wseg = em.W("seg_block.weight")
    seg_fused = [False]

    def tail_conv(x, ck, bk, last, out=None):
        """a decoder stage's last conv3x3+BN+ReLU; on the LAST active stage (eval) the seg_block head rides in its epilogue"""
        if (last and tp is None and out is None and p.can_conv_halo_seg(x, em.W(ck + ".weight").shape[0], wseg.shape[0])
                and not (DECODER_SPLITK and p.would_splitk(x.B * x.H * x.W, 9 * x.Cphys, em.W(ck + ".weight").shape[0]))):
            s_, t_ = em.ws.bn_fold(bk)
            seg_fused[0] = True
            return p.conv_halo_seg(x, ck, em.W(ck + ".weight"), s_, t_, ACT_RELU, "seg_block", wseg, em.W("seg_block.bias"), io["seg_tb"])
        return em.conv_bn(x, ck, bk, 3, 1, 1, out=out)

    for i in range(active):
        p.set_lane(0)
        up = "up_net.%d" % i
        last = i == active - 1
        if i == 0:   # ConvTranspose2d(k3,s2,p1,op1)+BN+ReLU as 4 sub-pixel phase convs, then 2x conv3x3+BN+ReLU
            wt = em.W(up + ".0.weight")                     # (Cin, Cout, 3, 3)
            s, t = em.ws.bn_fold(up + ".1") if tp is None else (None, None)
            o = p.act(2 * f.H, 2 * f.W, nf)
            if tp is not None:
                s, t = tp.const_vec(nf, True), tp.const_vec(nf, False)
            for ph in range(4):          # the four sub-pixel phases are independent: eval runs them on lanes 0, 3, 4, 5
                a, b = ph >> 1, ph & 1
                if tp is None and ph > 0:
                    p.set_lane(2 + ph)
                p.conv(f, up + ".0", wt, s, t, 1 + a, 1 + b, 1, 0, nf, ACT_RELU if tp is None else ACT_NONE, transposed=1, phase=ph,
                       ostr=((a * o.W + b) * o.cstride, o.H * o.W * o.cstride, 2 * o.W * o.cstride, 2 * o.cstride, 1),
                       out_tbuf=o.tbuf, out_hw=(f.H, f.W))
            if tp is None:
                p.set_lane(0)
                for ph in range(1, 4):
                    p.sync(2 + ph, 0)
            if tp is not None:
                o = _convt_train_tail(em, up, wt, f, o, nf)
            f = em.conv_bn(o, up + ".3", up + ".4", 3, 1, 1)
            f = tail_conv(f, up + ".6", up + ".7", last, out=lowcats[1].slice(0, nf) if 1 in lowcats else None)
        elif i in lowcats:   # the same, upsample fused into conv1's loader
            s1, t1 = em.ws.bn_fold(up + ".2")
            f = p.conv_up2x(lowcats[i], up + ".1", em.W(up + ".1.weight"), s1, t1, ACT_RELU)
            f = tail_conv(f, up + ".4", up + ".5", last, out=lowcats[i + 1].slice(0, nf) if (i + 1) in lowcats else None)
        else:        # cat[img_feat, img_feats[-i-1]] -> bilinear x2 (align_corners) -> 2x conv3x3+BN+ReLU
            sk = feats[-i - 1]
            cat = cats[i] if i in cats else p.act(2 * f.H, 2 * f.W, f.C + sk.C)
            if i == 1 and i in cats:
                p.sync(2, 0)             # all skip halves (lane 2) are in place
            p.upsample2x(f, cat.slice(0, f.C))
            if i not in cats:
                p.upsample2x(sk, cat.slice(f.Cphys, sk.C))
            if tp is not None:
                _upsample_tape(tp, f, cat.slice(0, f.C))
                _upsample_tape(tp, sk, cat.slice(f.Cphys, sk.C))
            f = em.conv_bn(cat, up + ".1", up + ".2", 3, 1, 1)
            f = tail_conv(f, up + ".4", up + ".5", last, out=lowcats[i + 1].slice(0, nf) if (i + 1) in lowcats else None)
        # ---- Refine_moduleGNN.forward pipeline.py:262-298
        p.sync(0, 1)
        p.set_lane(1)
        fdec = f                                                # the decoder's own map: what the next decoder stage continues from
        if tp is None and cfg.get("export_dec"):                # attribution hooks (HipForwardMixin._run(want_dec / inject_dec)): the map
            p.to_nchw_f32(f, io["dec_feats"][i])                # the refinement stage gathers from, exported / replaced by a GIVEN one
        if tp is None and cfg.get("inject_dec"):
            t_ = io["inject_dec"][i]
            f = p.nchw_to_nhwc(t_, t_.shape[1], t_.shape[2], t_.shape[3])
        rp = "refine_net.%d" % i
        wpg = em.W(rp + ".local_feat_ext_block.patch_generator.weight")     # (E, nf, k, k)
        Ech = wpg.shape[0]
        pgk = rp + ".local_feat_ext_block.patch_generator"
        if tp is None and p.can_gather_patch(f, N, Ech, k):     # conv only where it is gathered (the 64x64 stage at N = 512)
            p.index2feat_conv(f, rp + ".patch", wpg, em.W(pgk + ".bias"), io["xid"], io["yid"], io["mask"], L.slice(0, 4 * Ech), N, k)
            patches = None
        elif tp is None:
            patches = p.conv(f, rp + ".patch", wpg, em._unit(Ech), em.W(pgk + ".bias"), k, k, 1, k - 1, Ech, out_half=p.gnn_half)
        else:
            patches = p.conv(f, rp + ".patch", wpg, tp.const_vec(Ech, True), em._bias_vec(pgk, Ech), k, k, 1, k - 1, Ech)
        Lnext = local_buf(i + 1) if i + 1 < active else None
        # network_num_graph_module = 0 (config/lm/hr18GNN2_res6_gnn3Skip_mlpQuery_lm_woEdgeConv.txt): the pre-graph MLP's rows are the
        # stage's graph feature (pipeline.py:288-297 with an empty pre_query_block) -> they land in the next stage's input rows
        carry = Lnext.slice(qd[0], qd[0]) if (ngs[i] == 0 and Lnext is not None) else None
        carried = False
        pkeys = [rp + ".pre_graph_module.0", rp + ".pre_graph_module.2"]
        pws = [em.W(k_ + ".weight") for k_ in pkeys]
        gather_in_pair = tp is None and p.can_fuse_mlp_pair_gather(L, patches, pws[0], pws[1], Ech, k)
        if patches is not None and not gather_in_pair:
            p.index2feat(patches, io["xid"], io["yid"], io["mask"], L.slice(0, 4 * Ech), N, Ech, k)
        if tp is not None:
            # the decode ops advance io["xid"/"yid"] in place: keep this stage's gather positions for the backward
            ids_i = dict(mask=io["mask"])
            for nm in ("xid", "yid"):
                ids_i[nm] = torch.empty_like(io[nm])
                tp.memcpy(ids_i[nm].data_ptr(), io[nm].data_ptr(), io[nm].numel() * 4, "save_ids")
            _patch_tape(em, pgk, wpg, f, patches, L.slice(0, 4 * Ech), ids_i, N, Ech, k)
        if gather_in_pair:      # Index2Feat's gather x RoI bit inside the pair's loader: L's first 256 channels are never materialised
            h = p.mlp_pair_fused_gather(patches, io["xid"], io["yid"], io["mask"], L, pkeys, pws, [em.W(k_ + ".bias") for k_ in pkeys],
                                        slope, N, k)
        elif tp is None and p.can_fuse_mlp_pair(L, pws[0], pws[1]):
            h = p.mlp_pair_fused(L, pkeys, pws, [em.W(k_ + ".bias") for k_ in pkeys], slope)     # csrc/mlp_fused.hip
        else:
            hk = {"in_half": True} if (tp is None and p.gnn_half) else {}
            h = em.linear(L, rp + ".pre_graph_module.0", ACT_LEAKY, slope, **hk)
            h = em.linear(h, rp + ".pre_graph_module.2", ACT_LEAKY, slope, out=carry, **hk)
            carried = carry is not None
        if carry is not None and not carried:      # (the fused pair launches write a dense tensor) one term, no ReLU: cp_fuse_sum_act is a
            p.fuse_sum([h], [0], carry, relu=False)    # copy -- bit-exact for 16-bit rows of either format
        for gi in range(ngs[i]):
            last = gi == ngs[i] - 1
            h = em.edgeconv("%s.pre_query_block.%d" % (rp, gi), h, io["graph"], cfg["graph_slope"],
                            out=Lnext.slice(qd[0], qd[0]) if (last and Lnext is not None) else None)
        # Linear(64 -> 2): channel 0 = new x bit -> row 4+i, channel 1 = new y bit -> row 10+i  (pipeline.py:375-378)
        qk = rp + ".query_block.mlps.4"
        qkeys = [rp + ".query_block.mlps.%d" % j for j in (0, 2, 4)]
        qws = [em.W(k_ + ".weight") for k_ in qkeys]
        if tp is None and p.can_fuse_query_mlp(h, [qws[0].shape[1], qws[0].shape[0], qws[1].shape[0], qws[2].shape[0]]):
            # MLP_QueryNet as one launch: both hidden layers stay on chip, only the logits leave (csrc/mlp_fused.hip)
            p.mlp_query_fused(h, qkeys, qws, [em.W(k_ + ".bias") for k_ in qkeys], slope, io["bits_tb"],
                              ((4 + i) * N, 13 * N, 0, 1, 6 * N))
            q = None
        else:
            hk = {"in_half": True} if (tp is None and p.gnn_half) else {}
            q = em.linear(h, rp + ".query_block.mlps.0", ACT_LEAKY, slope, **hk)
            q = em.linear(q, rp + ".query_block.mlps.2", ACT_LEAKY, slope, **hk)
        if q is None:
            pass
        elif tp is None:
            em.linear(q, qk, ACT_NONE, 0.0, out_f32=True, ostr=((4 + i) * N, 13 * N, 0, 1, 6 * N), out_tbuf=io["bits_tb"], in_half=p.gnn_half)
        else:
            wq = em.W(qk + ".weight")
            wq4 = wq.view(wq.shape[0], wq.shape[1], 1, 1)
            p.conv(q, qk, wq4, tp.const_vec(2, True), em._bias_vec(qk, 2), 1, 1, 1, 0, 2, out_f32=True,
                   ostr=((4 + i) * N, 13 * N, 0, 1, 6 * N), out_tbuf=io["bits_tb"])
            em._out_layer_bwd(qk, wq4, q, 2, io["dbits"], (4 + i) * N, 13 * N, 1, 6 * N)
        p.decode(dbits, i, io["mask"], io["xid"], io["yid"], io["x64"], io["y64"], N)
        L = Lnext
        f = fdec
    # seg_block Conv2d(nf -> seg_dim, 1x1) on the last feature map, NCHW fp32 out (pipeline.py:349,383)
    if active > 0:
        p.set_lane(0)
    sd_ = wseg.shape[0]
    if seg_fused[0]:
        pass                                  # written by the last decoder conv's epilogue (engine.conv_halo_seg)
    elif tp is None:
        p.conv(f, "seg_block", wseg, em._unit(sd_), em.W("seg_block.bias"), 1, 1, 1, 0, sd_, out_f32=True,
               ostr=(0, sd_ * f.H * f.W, f.W, 1, f.H * f.W), out_tbuf=io["seg_tb"])
    else:
        p.conv(f, "seg_block", wseg, tp.const_vec(sd_, True), em._bias_vec("seg_block", sd_), 1, 1, 1, 0, sd_, out_f32=True,
               ostr=(0, sd_ * f.H * f.W, f.W, 1, f.H * f.W), out_tbuf=io["seg_tb"])
        em._out_layer_bwd("seg_block", wseg, f, sd_, io["dseg"], 0, sd_ * f.H * f.W, 1, f.H * f.W)
    if active > 0:
        p.par_end()
    return feats, f
