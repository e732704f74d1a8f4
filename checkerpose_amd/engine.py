"""Host-side engine: turns a module tree (state dict + config) into a static launch program over the C ABI.

Design: one process per GPU; a *program* is built once per (batch size, active stages, dtype):
every op's arguments (pointers into ONE liveness-planned workspace, ctypes descriptors) are resolved ahead of
time, so a forward is a flat loop of ~370 pre-bound C calls -- or one hipGraph replay of the captured loop.
Nothing here computes: torch is used for device memory, parameter access and the stream handle only, and every
arithmetic step is a HIP kernel behind include/checkerpose_hip.h.  No fallback path exists.
"""
import ctypes as C
import os

import torch

from . import _abi
from ._abi import ACT_LEAKY, ACT_NONE, ACT_RELU, CP_BF16, CP_F16, CP_F32, CpChainTail, CpChainTailConv, CpConvDesc, CpConvGroupItem, CpFuseConv

_TORCH_DT = {CP_F32: torch.float32, CP_BF16: torch.bfloat16}
USE_HALO = os.environ.get("CHECKERPOSE_AMD_HALO", "1") != "0"   # LDS-halo 3x3 kernel (A/B switch for kernel work)
USE_FUSED_BN = os.environ.get("CHECKERPOSE_AMD_FUSED_BN", "1") != "0"   # fused 256-64-64-256 Bottleneck kernel (bf16)
USE_FUSED_BB = os.environ.get("CHECKERPOSE_AMD_FUSED_BB", "1") != "0"   # fused BasicBlock kernel (C <= 32)
USE_GEMM = os.environ.get("CHECKERPOSE_AMD_GEMM", "1") != "0"   # LDS-staged 1x1 / Linear kernel
USE_CHAIN = os.environ.get("CHECKERPOSE_AMD_CHAIN", "1") != "0"   # one launch per HRNet branch chain (bf16, map resident in LDS)
USE_UP_FUSED = os.environ.get("CHECKERPOSE_AMD_UP_FUSED", "1") != "0"   # decoder: bilinear x2 interpolated inside the conv's halo loader
USE_FUSE_OUT = os.environ.get("CHECKERPOSE_AMD_FUSE_OUT", "1") != "0"   # HRNet fuse layers: first-level convs grouped by source branch
USE_S2_SMALL = os.environ.get("CHECKERPOSE_AMD_S2_SMALL", "1") != "0"   # LDS-staged 3x3 / stride-2 conv for wide inputs (transition1[1])
USE_SEG_FUSED = os.environ.get("CHECKERPOSE_AMD_SEG_FUSED", "1") != "0"   # seg_block inside the last decoder conv's epilogue
USE_HALO2 = os.environ.get("CHECKERPOSE_AMD_HALO2", "1") != "0"     # k = 2 / pad 1 convs on the LDS-staged halo kernel (A/B: the generic kernel)
USE_CHAIN_TAILS = os.environ.get("CHECKERPOSE_AMD_CHAIN_TAILS", "1") != "0"   # the 36 / 72 / 144-channel chain launches also run the first-level fuse convs fed by their branch (A/B switch)
USE_CHAIN_TAIL = os.environ.get("CHECKERPOSE_AMD_CHAIN_TAIL", "1") != "0"   # the 64x64 chain launch also runs the stride-2 fuse convs that read its output
FUSE_OUT_MIN_BATCH = int(os.environ.get("CHECKERPOSE_AMD_FUSE_OUT_MIN_BATCH", "1"))   # grouped first-level fuse-layer launches: at every batch
#   (after per-conv branches too; 350 -> 317 graph nodes below 40 crops: B = 1 1.60 -> 1.58 ms, B = 8 1.88 -> 1.68, B = 32 3.10 -> 2.99; -1: with the chains)
USE_MLP_FUSED = os.environ.get("CHECKERPOSE_AMD_MLP_FUSED", "1") != "0"   # MLP_QueryNet's three Linears as one launch (bf16)
# (no lower bound: measured with the fused launches at every batch size against a 32 768-row threshold, ms per forward: B = 1 1.60 vs 1.77,
#  B = 8 1.84 vs 1.94, B = 32 3.11 vs 3.17 -- below 64 crops a forward is bound by its ~350 dependent graph nodes, and the stacks are 9 fewer)
MLP_FUSED_MIN_ROWS = int(os.environ.get("CHECKERPOSE_AMD_MLP_FUSED_MIN_ROWS", "1"))
EDGE_SCHED = os.environ.get("CHECKERPOSE_AMD_EDGE_SCHED", "1") != "0"   # A/B: bank-conflict-aware neighbour order for edge_fused
USE_CONV_GROUP = os.environ.get("CHECKERPOSE_AMD_CONV_GROUP", "0") == "1"    # training: independent small 3x3 convs in one launch (measured: no gain)
CONV_GROUP_MAX_C = int(os.environ.get("CHECKERPOSE_AMD_CONV_GROUP_MAXC", "48"))
USE_SPLITK = os.environ.get("CHECKERPOSE_AMD_SPLITK", "1") != "0"     # small-batch split-K routing (cp_conv2d_igemm_splitk)
GEMM_WS_SMALL_K = os.environ.get("CHECKERPOSE_AMD_GEMM_WS_SMALL_K", "1") != "0"   # A/B: weight-stationary GEMM from K = 64
USE_MLP_GATHER = os.environ.get("CHECKERPOSE_AMD_MLP_GATHER", "1") != "0"   # Index2Feat's 4-tap gather inside the fused MLP pair's loader (A/B switch)
USE_PATCH_GATHER = os.environ.get("CHECKERPOSE_AMD_PATCH_GATHER", "1") != "0"   # patch conv only at the gathered taps
USE_STEM = os.environ.get("CHECKERPOSE_AMD_STEM", "1") != "0"   # fused HRNet stem (bf16)
STEM_MIN_BATCH = int(os.environ.get("CHECKERPOSE_AMD_STEM_MIN_BATCH", "96"))   # one persistent workgroup per crop: measured
#   crossover against the tiled launches (tools/batch_sweep.py, MI355X): chains win from 64 crops, stem + EdgeConv from 96
USE_EDGE_FUSED = os.environ.get("CHECKERPOSE_AMD_EDGE_FUSED", "1") != "0"   # EdgeConv layer (node GEMM + gather-max) in one launch
# bf16 programs run the per-keypoint (GNN) block group -- conv1x1's output rows, the EdgeConv layers, Index2Feat's rows, both MLP stacks --
# in IEEE half (CP_F16: same bytes, same MFMA rate, 3 more mantissa bits) when every op of the group takes its fused per-crop kernel;
# on trained-like weights that group produced 73 % of the bf16 path's logit-error variance.  A/B switch.
USE_GNN_F16 = os.environ.get("CHECKERPOSE_AMD_GNN_F16", "1") != "0"
USE_EDGE_TILED = os.environ.get("CHECKERPOSE_AMD_EDGE_TILED", "1") != "0"   # N > 512: patches of 512 keypoints, table slices in LDS (cp_edgeconv_tiled)
EDGE_FUSED_MIN_BATCH = int(os.environ.get("CHECKERPOSE_AMD_EDGE_FUSED_MIN_BATCH", "96"))   # one workgroup per crop: needs crops to fill the chip
CHAIN_MIN_BATCH = int(os.environ.get("CHECKERPOSE_AMD_CHAIN_MIN_BATCH", "40"))   # below: per-conv launches (a crop's chain runs on ONE CU)
NO_RECYCLE = os.environ.get("CHECKERPOSE_AMD_NO_RECYCLE", "0") == "1"   # debugging aid: every workspace tensor gets its own bytes
DTYPES = {"fp32": CP_F32, "f32": CP_F32, "float32": CP_F32, "bf16": CP_BF16, "bfloat16": CP_BF16}


class _ConvLog(list):
    """conv_log entries remember which launch they describe (dead-launch elimination drops both together)"""

    def __init__(self, prog):
        super().__init__()
        self.prog, self.op_index = prog, []

    def append(self, e):
        super().append(e)
        self.op_index.append(len(self.prog.ops) - 1)


def _rup(x, m):
    return (x + m - 1) // m * m


class TBuf:
    """A workspace tensor: `nbytes` at `offset` of the program's workspace; live over ops [first, last]."""
    __slots__ = ("nbytes", "offset", "first", "last", "fixed")

    def __init__(self, nbytes):
        self.nbytes, self.offset, self.first, self.last, self.fixed = _rup(nbytes, 256), None, None, None, None


class Act:
    """Channels-last activation view: channels [coff, coff+Cphys) of a (B, H, W, cstride) tensor in `tbuf`."""
    __slots__ = ("tbuf", "B", "H", "W", "C", "Cphys", "cstride", "coff")

    def __init__(self, tbuf, B, H, W, C, Cphys, cstride, coff):
        self.tbuf, self.B, self.H, self.W, self.C, self.Cphys, self.cstride, self.coff = tbuf, B, H, W, C, Cphys, cstride, coff

    def slice(self, coff, C, Cphys=None):
        return Act(self.tbuf, self.B, self.H, self.W, C, Cphys if Cphys is not None else C, self.cstride, self.coff + coff)


class WeightStore:
    """Packed (MFMA-fragment order) weights + folded per-channel affine vectors, built from a device state dict."""

    def __init__(self, lib, sd, dtype, device):
        self.lib, self.sd, self.dtype, self.device = lib, sd, dtype, device
        self.E = lib.cp_chan_align(dtype)
        self.cache = {}
        self.keep = []   # keep temporaries alive until packing kernels ran
        self.repacks_every_step = False      # trainer.TrainWeightStore: True

    def _w(self, key):
        t = self.sd[key]
        if t.device != self.device or t.dtype != torch.float32:
            raise RuntimeError("parameter %s must be an fp32 tensor on %s" % (key, self.device))
        return t.detach().contiguous()

    def pack(self, name, w, Cout, Cin, R, S, cin_phys, cout_rows, transposed=0, phase=0, row_map=None, dtype=None):
        dtype = self.dtype if dtype is None else dtype
        ck = (name, cin_phys, cout_rows, transposed, phase) + (() if dtype == self.dtype else (dtype,))
        if ck in self.cache:
            return self.cache[ck]
        nbytes = self.lib.cp_packed_weight_bytes(dtype, cout_rows, cin_phys, R, S)
        out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        rm = None
        if row_map is not None:
            rm = torch.tensor(row_map, dtype=torch.int32, device=self.device)
            self.keep.append(rm)
        w = w.contiguous()
        self.keep.append(w)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _abi.check(self.lib.cp_pack_conv_weight(st, dtype, w.data_ptr(), Cout, Cin, R, S, cin_phys, transposed, phase,
                                                rm.data_ptr() if rm is not None else None, cout_rows, out.data_ptr()),
                   "cp_pack_conv_weight(%s)" % name)
        self.cache[ck] = out
        return out

    def pack_halo(self, name, w, Cout, Cin, cin_phys):
        """3x3/s1/p1 weights in the halo kernel's image (cp_pack_conv3x3_halo_weight)."""
        ck = ("halo", name, cin_phys)
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_halo_weight_bytes(self.dtype, Cout, cin_phys), dtype=torch.uint8, device=self.device)
        w = w.contiguous()
        self.keep.append(w)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _abi.check(self.lib.cp_pack_conv3x3_halo_weight(st, self.dtype, w.data_ptr(), Cout, Cin, cin_phys, out.data_ptr()),
                   "cp_pack_conv3x3_halo_weight(%s)" % name)
        self.cache[ck] = out
        return out

    def pack_rows3x3(self, name, w, Cout, Cin, cin_phys):
        """conv1 of the fused BasicBlock (cp_pack_conv3x3_rows_weight: small-Cout halo image, rows unpermuted)."""
        ck = ("rows3x3", name, cin_phys)
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_halo_weight_bytes(self.dtype, Cout, cin_phys), dtype=torch.uint8, device=self.device)
        w = w.contiguous()
        self.keep.append(w)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _abi.check(self.lib.cp_pack_conv3x3_rows_weight(st, self.dtype, w.data_ptr(), Cout, Cin, cin_phys, out.data_ptr()),
                   "cp_pack_conv3x3_rows_weight(%s)" % name)
        self.cache[ck] = out
        return out

    def pack_gemm(self, name, w, Cout, Cin, cin_phys, dtype=None):
        """1x1 / Linear weights in the LDS-GEMM kernel's image (cp_pack_gemm_weight)."""
        dtype = self.dtype if dtype is None else dtype
        ck = ("gemm", name, cin_phys) + (() if dtype == self.dtype else (dtype,))
        if ck in self.cache:
            return self.cache[ck]
        out = torch.empty(self.lib.cp_packed_gemm_weight_bytes(dtype, Cout, cin_phys), dtype=torch.uint8, device=self.device)
        w = w.contiguous()
        self.keep.append(w)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _abi.check(self.lib.cp_pack_gemm_weight(st, dtype, w.data_ptr(), Cout, Cin, cin_phys, out.data_ptr()),
                   "cp_pack_gemm_weight(%s)" % name)
        self.cache[ck] = out
        return out

    def pack_chain(self, name, ws, affs, C_, H, W):
        """the 8 convs of an HRNet branch chain: one packed weight blob (BN scale folded in) + one [8][2][AFF] fp32 affine tensor
        (row 0: the scale, informational -- the kernel reads row 1, the shift)"""
        ck = ("chain", name)
        if ck in self.cache:
            return self.cache[ck]
        blob = torch.empty(self.lib.cp_hr_chain_weight_bytes(C_, H, W), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        for i, w in enumerate(ws):
            w = w.contiguous()
            sc = affs[i][0].to(device=self.device, dtype=torch.float32).contiguous()      # folded-BN scale goes INTO the packed weights
            self.keep += [w, sc]
            _abi.check(self.lib.cp_pack_hr_chain_weight(st, w.data_ptr(), sc.data_ptr(), C_, H, W, i, blob.data_ptr()),
                       "cp_pack_hr_chain_weight(%s)" % name)
        n = self.lib.cp_hr_chain_affine_floats(C_, H, W)
        aff = torch.zeros(8, 2, n, dtype=torch.float32, device=self.device)
        for i, (s_, t_) in enumerate(affs):
            aff[i, 0, :C_] = s_
            aff[i, 1, :C_] = t_
        self.cache[ck] = (blob, aff)
        return self.cache[ck]

    def affine(self, name, scale, shift, rows):
        """fp32 scale/shift vectors padded with zeros to a multiple of 16 entries (rows >= len)."""
        ck = ("aff", name, rows)
        if ck not in self.cache:
            n = _rup(rows, 16)
            s = torch.zeros(n, dtype=torch.float32, device=self.device)
            t = torch.zeros(n, dtype=torch.float32, device=self.device)
            s[: scale.numel()] = scale
            t[: shift.numel()] = shift
            self.cache[ck] = (s, t)
        return self.cache[ck]

    def bn_fold(self, bn, eps=1e-5):
        """eval BatchNorm -> (scale, shift): y = x*scale + shift (nn.BatchNorm2d defaults, eps 1e-5)."""
        s = self._w(bn + ".weight") / torch.sqrt(self._w(bn + ".running_var") + eps)
        return s, self._w(bn + ".bias") - self._w(bn + ".running_mean") * s


class Program:
    """Static launch list for one (B, dtype).  Build with the op methods, then finalize(); run() replays it."""

    def __init__(self, lib, ws: WeightStore, dtype, B, device):
        self.lib, self.ws, self.dtype, self.B, self.device = lib, ws, dtype, B, device
        self.E = lib.cp_chan_align(dtype)
        self.es = 2 if dtype == CP_BF16 else 4
        self.ops = []          # (fn, argbuilder(ptr_of) -> args tuple, name, reads, writes)
        self.tbufs = []
        self.keep = []
        self.workspace = None
        self.calls = None
        self.lane = 0          # current launch lane (stream) -- see par_begin()
        self.nlanes = 1
        self.regions = []      # [start_op, end_op] of fork/join regions (buffer lifetimes are extended over them)
        self._open = None
        self.flops = 0         # dense MACs*2 issued through cp_conv2d_igemm (algorithmic, unpadded)
        self.conv_log = _ConvLog(self)     # (name, M, Cout, K, flops, family, bytes) per MFMA launch -- bench roofline uses it
        self._rw = {}          # op index -> (reads, writes): dead-launch elimination in finalize(dce=True), hazards of run_dag
        # launch selection by batch size (module globals = the measured crossovers); HipForwardMixin.set_kernel_selection pins one
        # selection for every batch size, so a crop's bf16 bits do not depend on the size of the batch it arrives in
        self.chain_min, self.stem_min, self.edge_min, self.splitk = CHAIN_MIN_BATCH, STEM_MIN_BATCH, EDGE_FUSED_MIN_BATCH, USE_SPLITK
        self.mlp_min_rows = MLP_FUSED_MIN_ROWS
        self.fuse_out_min = FUSE_OUT_MIN_BATCH if FUSE_OUT_MIN_BATCH >= 0 else CHAIN_MIN_BATCH
        self._raw = {}         # (ptr, nbytes) -> TBuf of a raw-pointer operand (see raw())
        self.gnn_half = False  # the keypoint-side tensors of this (bf16) program are IEEE half (set by the runtime, see USE_GNN_F16)

    @property
    def gdt(self):
        """storage / MFMA type of the keypoint-side (GNN) block group"""
        return CP_F16 if self.gnn_half else self.dtype

    def wants_gnn_half(self, N, K, tiled_hpad=None):
        """bf16 programs run their keypoint side in half at every batch size and kernel selection: the fused per-crop kernels and the
        generic ones (cp_conv2d_igemm incl. split-K, cp_gemm_rows, cp_edgeconv_gather_max, cp_index2feat_gather) all take CP_F16"""
        return bool(USE_GNN_F16 and self.dtype == CP_BF16)

    # ---- tensors
    def tensor(self, nelem, es=None):
        t = TBuf(nelem * (es or self.es))
        self.tbufs.append(t)
        return t

    def act(self, H, W, C, cstride_C=None):
        Cphys = _rup(C, self.E)
        cs = Cphys if cstride_C is None else cstride_C
        return Act(self.tensor(self.B * H * W * cs), self.B, H, W, C, Cphys, cs, 0)

    def fixed(self, torch_tensor):
        t = TBuf(torch_tensor.numel() * torch_tensor.element_size())
        t.fixed = torch_tensor
        self.keep.append(torch_tensor)
        return t

    def raw(self, torch_tensor):
        """The TBuf standing for a caller-owned tensor an op touches through a raw pointer (index / mask / logits buffers):
        listed in the op's reads / writes so the dataflow capture (run_dag) orders its producer and consumers."""
        key = (torch_tensor.data_ptr(), torch_tensor.numel() * torch_tensor.element_size())
        if key not in self._raw:
            t = TBuf(key[1])
            t.nbytes = key[1]
            t.fixed = torch_tensor
            self._raw[key] = t
        return self._raw[key]

    def _add(self, fn, argb, name, reads, writes):
        i = len(self.ops)
        for t in list(reads) + list(writes):
            if t is None or t.fixed is not None:
                continue
            if t.first is None:
                t.first = i
            t.last = i
        self._rw[i] = ([t for t in reads if t is not None], [t for t in writes if t is not None])
        self.ops.append((fn, argb, name, self.lane))

    # ---- structured concurrency: independent sub-chains (HRNet branches, decoder vs GNN refinement) are launched on
    # separate lanes; under hipGraph capture each lane is its own stream joined by events, so the graph gets parallel
    # branches and the latency-bound small kernels overlap the big ones.  Eager replay ignores lanes (one stream,
    # program order) and is therefore the sequential reference the captured graph must reproduce bit for bit.
    def par_begin(self, nlanes):
        assert self._open is None, "regions do not nest"
        self.ops.append(("__fork__", nlanes))
        self._open = len(self.ops)
        self.nlanes = max(self.nlanes, nlanes)
        self.lane = 0

    def set_lane(self, k):
        assert self._open is not None
        self.lane = k

    def sync(self, src, dst):
        """lane `dst` waits for everything launched on lane `src` so far"""
        assert self._open is not None and src != dst
        self.ops.append(("__sync__", src, dst))

    def mark(self, lane):
        """remember the current tail of `lane`; wait(token, dst) later makes lane `dst` wait for exactly that point"""
        assert self._open is not None
        self._nmarks = getattr(self, "_nmarks", 0) + 1
        self.ops.append(("__mark__", self._nmarks, lane))
        return self._nmarks

    def wait(self, token, dst):
        assert self._open is not None
        self.ops.append(("__wait__", token, dst))

    def par_end(self):
        assert self._open is not None
        self.ops.append(("__join__",))
        self.regions.append((self._open - 1, len(self.ops) - 1))
        self._open = None
        self.lane = 0

    # ---- ops
    def conv(self, x: Act, wkey, w, scale, shift, R, S, stride, pad, Cout, act=ACT_NONE, slope=0.0, residual: Act = None,
             out: Act = None, transposed=0, phase=0, row_map=None, cout_rows=None, out_f32=False, ostr=None,
             out_tbuf=None, out_hw=None, w_shape=None, out_half=False, in_half=False):
        """Generic conv / linear.  `w` is the fp32 source weight in PyTorch layout; returns the output Act.
        out_half: a bf16 conv whose output rows are IEEE half (the producer of a keypoint-side tensor, CpConvDesc.out_f32 = 2);
        in_half: input rows and weights are IEEE half (a keypoint-side Linear on the generic kernel, CpConvDesc.dtype = CP_F16)."""
        if out_half or in_half:
            assert self.dtype == CP_BF16 and residual is None and not (out_half and out_f32)
        E = self.E
        wCout, wCin = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
        rows = cout_rows if cout_rows is not None else wCout
        if wCin != x.C:
            raise RuntimeError("conv %s: weight expects %d input channels, activation has %d" % (wkey, wCin, x.C))
        halo = (USE_HALO and R == 3 and S == 3 and stride == 1 and pad == 1 and ostr is None and not out_f32
                and not transposed and row_map is None and x.W >= 16 and x.H >= 8 and not (out_half or in_half))
        cdt = CP_F16 if in_half else self.dtype          # the launch's dtype: half rows + half weights for a keypoint-side layer
        if halo and wCout <= 80 and self.dtype == CP_F32:
            # small-Cout variant pads Cin to 64-byte chunks PER TAP: in fp32 (MFMA-bound) that only pays when the
            # padding waste is small (measured: 18/36-channel convs are faster on the generic kernel in fp32)
            halo = _rup(x.Cphys, 16) <= 1.15 * x.Cphys
        # K >= 4 chunks; from 16384 rows the bf16 weight-stationary variant runs (weights in registers, rows streamed once), which
        # also pays at K = 64 .. 127 (incre conv3 + shortcut of the 32^2 branch: 98 -> 72 us)
        kmin = 16 * self.E if not (self.dtype == CP_BF16 and residual is None and x.B * x.H * x.W >= 16384 and GEMM_WS_SMALL_K) else 64
        gemm = (USE_GEMM and R == 1 and S == 1 and stride == 1 and pad == 0 and ostr is None and not out_f32
                and not transposed and row_map is None and wCout >= 96 and x.Cphys >= kmin and not out_half)
        s2small = (USE_S2_SMALL and self.dtype == CP_BF16 and R == 3 and S == 3 and stride == 2 and pad == 1 and ostr is None and not out_f32
                   and not transposed and row_map is None and residual is None and x.H % 2 == 0 and x.W % 2 == 0 and x.B >= self.chain_min
                   and not (out_half or in_half) and bool(self.lib.cp_conv3x3_s2_small_supported(x.H, x.W, x.Cphys, _rup(wCout, self.E))))
        # k = 2 / pad 1 (Index2Feat's patch_generator over the whole map): the small-Cout halo kernel with four taps
        halo2 = (USE_HALO2 and self.dtype == CP_BF16 and R == 2 and S == 2 and stride == 1 and pad == 1 and ostr is None and not out_f32
                 and not transposed and row_map is None and not in_half and not self.ws.repacks_every_step   # (eval programs:
                 # the training program repacks its weights every step through the pack-item tables, which have no entry for this image)
                 and bool(self.lib.cp_conv2x2_halo_supported(self.dtype, x.H, x.W, _rup(wCout, self.E))))
        if (self.splitk and not transposed and row_map is None and (halo or gemm or s2small or halo2)
                and self.lib.cp_conv2d_igemm_splitk(cdt, x.B * ((x.H + 2 * pad - R) // stride + 1) * ((x.W + 2 * pad - S) // stride + 1),
                                                    R * S * x.Cphys, _rup(wCout, self.E))):
            halo = gemm = s2small = halo2 = False      # small batch: the generic kernel's split-K variant beats the tiled specialists
        if s2small:
            ck = ("s2small", wkey, x.Cphys)
            if ck not in self.ws.cache:
                buf = torch.empty(self.lib.cp_conv3x3_s2_small_weight_bytes(x.Cphys, _rup(wCout, self.E)), dtype=torch.uint8, device=self.device)
                wc = w.contiguous()
                self.ws.keep.append(wc)
                st_ = torch.cuda.current_stream(self.device).cuda_stream
                _abi.check(self.lib.cp_pack_conv3x3_s2_small_weight(st_, wc.data_ptr(), wCout, wCin, x.Cphys, _rup(wCout, self.E), buf.data_ptr()),
                           "cp_pack_conv3x3_s2_small_weight(%s)" % wkey)
                self.ws.cache[ck] = buf
            packed = self.ws.cache[ck]
            halo = gemm = False
        elif halo2:
            ck = ("halo2", wkey, x.Cphys)
            if ck not in self.ws.cache:
                buf = torch.empty(self.lib.cp_packed_conv2x2_halo_weight_bytes(self.dtype, wCout, x.Cphys), dtype=torch.uint8, device=self.device)
                wc = w.contiguous()
                self.ws.keep.append(wc)
                st_ = torch.cuda.current_stream(self.device).cuda_stream
                _abi.check(self.lib.cp_pack_conv2x2_halo_weight(st_, self.dtype, wc.data_ptr(), wCout, wCin, x.Cphys, buf.data_ptr()),
                           "cp_pack_conv2x2_halo_weight(%s)" % wkey)
                self.ws.cache[ck] = buf
            packed = self.ws.cache[ck]
        elif halo:
            packed = self.ws.pack_halo(wkey, w, wCout, wCin, x.Cphys)
        elif gemm:
            packed = self.ws.pack_gemm(wkey, w, wCout, wCin, x.Cphys, **({"dtype": CP_F16} if in_half else {}))
        else:
            packed = self.ws.pack(wkey, w, wCout, wCin, R, S, x.Cphys, rows, transposed, phase, row_map, **({"dtype": CP_F16} if in_half else {}))
        sc, sh = self.ws.affine(wkey + "#" + str(phase), scale, shift, rows)
        Ho, Wo = out_hw if out_hw is not None else ((x.H + 2 * pad - R) // stride + 1, (x.W + 2 * pad - S) // stride + 1)
        d = CpConvDesc()
        d.dtype, d.out_f32 = (CP_F16 if in_half else self.dtype), (1 if out_f32 else (2 if out_half else 0))
        d.B, d.H, d.W = x.B, x.H, x.W
        d.Cin, d.in_cstride, d.in_coff = x.Cphys, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = R, S, stride, pad, Ho, Wo
        d.act, d.slope = act, slope
        d.ksplit = 0 if self.splitk else -1
        if ostr is None:
            if out is None:
                out = self.act(Ho, Wo, Cout)
            d.Cout = out.Cphys
            d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
            otb = out.tbuf
        else:  # explicit element strides into `out_tbuf` (o_base, o_sb, o_sy, o_sx, o_sc)
            d.Cout = Cout
            d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = ostr
            otb = out_tbuf
        self.keep += [d, packed, sc, sh]
        rtb = residual.tbuf if residual is not None else None
        if residual is not None and ostr is None:
            assert (residual.cstride, residual.coff, residual.H, residual.W) == (out.cstride, out.coff, out.H, out.W), \
                "residual must share the output layout"
        fn = self.lib.cp_conv3x3_halo if halo else (self.lib.cp_gemm_rows if gemm else self.lib.cp_conv2d_igemm)
        if halo2:
            fn = self.lib.cp_conv2x2_halo
        dref = C.byref(d)
        pw, ps, pt = packed.data_ptr(), sc.data_ptr(), sh.data_ptr()
        xtb = x.tbuf
        if halo:
            fam = "conv3x3_halo_s" if wCout <= 80 else ("conv3x3_halo4" if wCout % 256 == 0 else "conv3x3_halo")
        else:
            fam = "gemm_rows" if gemm else "conv_igemm"
        if halo2:
            fam = "conv3x3_halo_s"
        if s2small:
            fam = "conv3x3_s2_small"
            self._add(self.lib.cp_conv3x3_s2_small, lambda P: (dref, P(xtb), pw, ps, pt, P(otb)), fam + ":" + wkey, [xtb], [otb])
        else:
            self._add(fn, lambda P: (dref, P(xtb), pw, ps, pt, P(rtb) if rtb is not None else None, P(otb)),
                      fam + ":" + wkey, [xtb, rtb], [otb])
        fl = 2 * x.B * Ho * Wo * R * S * wCin * wCout
        self.flops += fl
        oes = 4 if out_f32 else self.es
        nbytes = (x.B * x.H * x.W * wCin * self.es + x.B * Ho * Wo * wCout * oes * (2 if residual is not None else 1)
                  + R * S * wCin * wCout * self.es)          # algorithmic: input + output (+ residual) + weights, unpadded
        self.conv_log.append((wkey, x.B * Ho * Wo, wCout, R * S * wCin, fl, fam, nbytes))
        return out

    def conv3x3_group(self, members):
        """INDEPENDENT 3x3 / stride 1 / pad 1 convs (the branches of an HRNet module at equal depth) -> one cp_conv3x3_halo_group launch
        for those the grouped kernel supports (<= 80 output channels, map >= 8 x 16), plain conv() launches for the rest.
        members: [(x, wkey, w, scale, shift, act, slope, residual, out)]; no two members may write the same tensor.  -> [out Act]"""
        lib = self.lib
        outs, grp = [None] * len(members), []
        for i, (x, wkey, w, scale, shift, act, slope, residual, out) in enumerate(members):
            wCout, wCin = w.shape[0], w.shape[1]
            if wCin != x.C:
                raise RuntimeError("conv %s: weight expects %d input channels, activation has %d" % (wkey, wCin, x.C))
            if USE_CONV_GROUP and _rup(wCout, self.E) <= CONV_GROUP_MAX_C and lib.cp_conv3x3_halo_group_supported(self.dtype, x.H, x.W, _rup(wCout, self.E)):
                grp.append(i)
            else:
                outs[i] = self.conv(x, wkey, w, scale, shift, 3, 3, 1, 1, wCout, act, slope, residual=residual, out=out)
        if len(grp) == 1:
            x, wkey, w, scale, shift, act, slope, residual, out = members[grp[0]]
            outs[grp[0]] = self.conv(x, wkey, w, scale, shift, 3, 3, 1, 1, w.shape[0], act, slope, residual=residual, out=out)
            grp = []
        if not grp:
            return outs
        builders, reads, writes, names = [], [], [], []
        for i in grp:
            x, wkey, w, scale, shift, act, slope, residual, out = members[i]
            wCout, wCin = w.shape[0], w.shape[1]
            packed = self.ws.pack_halo(wkey, w, wCout, wCin, x.Cphys)
            sc, sh = self.ws.affine(wkey + "#0", scale, shift, wCout)
            if out is None:
                out = self.act(x.H, x.W, wCout)
            d = CpConvDesc()
            d.dtype, d.out_f32 = self.dtype, 0
            d.B, d.H, d.W = x.B, x.H, x.W
            d.Cin, d.in_cstride, d.in_coff = x.Cphys, x.cstride, x.coff
            d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 3, 3, 1, 1, x.H, x.W
            d.act, d.slope, d.ksplit = act, slope, -1
            d.Cout = out.Cphys
            d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
            if residual is not None:
                assert (residual.cstride, residual.coff, residual.H, residual.W) == (out.cstride, out.coff, out.H, out.W), \
                    "residual must share the output layout"
            self.keep += [d, packed, sc, sh]
            rtb = residual.tbuf if residual is not None else None

            def build(P, d=d, xtb=x.tbuf, rtb=rtb, otb=out.tbuf, pw=packed.data_ptr(), ps=sc.data_ptr(), pt=sh.data_ptr(), wkey=wkey):
                it = CpConvGroupItem()
                _abi.check(lib.cp_conv3x3_halo_item(C.byref(d), P(xtb), pw, ps, pt, P(rtb) if rtb is not None else None, P(otb), C.byref(it)),
                           "cp_conv3x3_halo_item(%s)" % wkey)
                return it
            builders.append(build)
            reads += [x.tbuf] + ([rtb] if rtb is not None else [])
            writes.append(out.tbuf)
            names.append(wkey)
            outs[i] = out
            fl = 2 * x.B * x.H * x.W * 9 * wCin * wCout
            self.flops += fl
            nbytes = (x.B * x.H * x.W * (wCin + wCout * (2 if residual is not None else 1)) + 9 * wCin * wCout) * self.es
            self.conv_log.append((wkey, x.B * x.H * x.W, wCout, 9 * wCin, fl, "conv3x3_halo_group", nbytes))
        assert len(set(id(t) for t in writes)) == len(writes), "conv3x3_group: two members write the same tensor"
        dt = self.dtype

        def argb(P):
            items = [b(P) for b in builders]
            raw, prefix, total = _abi.device_table(items, [it.blocks for it in items], self.device)
            self.keep += [raw, prefix]
            return (dt, raw.data_ptr(), prefix.data_ptr(), len(items), total, max(int(it.lds_bytes) for it in items))
        self._add(lib.cp_conv3x3_halo_group, argb, "conv3x3_halo_group:" + names[0] + "+%d" % (len(names) - 1), reads, writes)
        return outs

    def would_splitk(self, M, K, Cout):
        """small-batch regime: cp_conv2d_igemm would run this conv as its split-K variant (M output pixels, K = R*S*Cin physical)"""
        return bool(self.splitk and self.lib.cp_conv2d_igemm_splitk(self.dtype, M, K, _rup(Cout, self.E)))

    def can_conv_up2x(self, H, W, Cout):
        """(H, W): the low-resolution input; the conv runs at (2H, 2W), which must tile like the plain halo kernel's maps"""
        return (USE_UP_FUSED and 2 * H >= 8 and 2 * W >= 16 and
                bool(self.lib.cp_conv3x3_halo_up2x_supported(self.dtype, _rup(Cout, self.E))))

    def conv_up2x(self, x: Act, wkey, w, scale, shift, act=ACT_NONE, slope=0.0, out: Act = None):
        """conv3x3(bilinear_x2(x)) in one launch (cp_conv3x3_halo_up2x): x is the LOW-resolution input, possibly a channel
        slice; the upsampled tensor never exists."""
        wCout, wCin = w.shape[0], w.shape[1]
        if wCin != x.C:
            raise RuntimeError("conv %s: weight expects %d input channels, activation has %d" % (wkey, wCin, x.C))
        packed = self.ws.pack_halo(wkey, w, wCout, wCin, x.Cphys)
        sc, sh = self.ws.affine(wkey + "#0", scale, shift, wCout)
        Ho, Wo = 2 * x.H, 2 * x.W
        if out is None:
            out = self.act(Ho, Wo, wCout)
        d = CpConvDesc()
        d.dtype, d.out_f32 = self.dtype, 0
        d.B, d.H, d.W = x.B, Ho, Wo
        d.Cin, d.in_cstride, d.in_coff = x.Cphys, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 3, 3, 1, 1, Ho, Wo
        d.act, d.slope, d.Cout = act, slope, out.Cphys
        d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
        self.keep += [d, packed, sc, sh]
        fn = self.lib.cp_conv3x3_halo_up2x
        dref = C.byref(d)
        pw, ps, pt = packed.data_ptr(), sc.data_ptr(), sh.data_ptr()
        xtb, otb = x.tbuf, out.tbuf
        self._add(fn, lambda P: (dref, P(xtb), pw, ps, pt, P(otb)), "conv3x3_halo4:" + wkey, [xtb], [otb])
        fl = 2 * x.B * Ho * Wo * 9 * wCin * wCout
        self.flops += fl
        nbytes = x.B * x.H * x.W * wCin * self.es + x.B * Ho * Wo * wCout * self.es + 9 * wCin * wCout * self.es
        self.conv_log.append((wkey, x.B * Ho * Wo, wCout, 9 * wCin, fl, "conv3x3_halo4", nbytes))
        return out

    def can_conv_halo_seg(self, x: Act, Cout, S):
        """3x3/s1/p1 conv + the 1x1 head on its output in one launch (cp_conv3x3_halo_seg): Cout == 256, S <= 2 head outputs"""
        return (USE_SEG_FUSED and USE_HALO and x.W >= 16 and x.H >= 8 and
                bool(self.lib.cp_conv3x3_halo_seg_supported(self.dtype, _rup(Cout, self.E), S)))

    def conv_halo_seg(self, x: Act, wkey, w, scale, shift, act, seg_key, seg_w, seg_b, seg_tbuf, slope=0.0):
        """act(bn(conv3x3(x))) -> Act, and seg = Conv2d(Cout -> S, 1x1)(that) + bias into the caller's (B, S, H, W) fp32 tensor"""
        wCout, wCin = w.shape[0], w.shape[1]
        if wCin != x.C:
            raise RuntimeError("conv %s: weight expects %d input channels, activation has %d" % (wkey, wCin, x.C))
        packed = self.ws.pack_halo(wkey, w, wCout, wCin, x.Cphys)
        sc, sh = self.ws.affine(wkey + "#0", scale, shift, wCout)
        S = seg_w.shape[0]
        ck = ("seg_head", seg_key, self.dtype)
        if ck not in self.ws.cache:
            w2 = seg_w.reshape(S, wCout).float()
            if self.dtype == CP_BF16:
                w2 = w2.to(torch.bfloat16).float()          # what the packed bf16 weights of the stand-alone head conv hold
            self.ws.cache[ck] = (w2.contiguous().to(self.device), seg_b.float().contiguous().to(self.device))
        sw, sb = self.ws.cache[ck]
        out = self.act(x.H, x.W, wCout)
        d = CpConvDesc()
        d.dtype, d.out_f32 = self.dtype, 0
        d.B, d.H, d.W = x.B, x.H, x.W
        d.Cin, d.in_cstride, d.in_coff = x.Cphys, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 3, 3, 1, 1, x.H, x.W
        d.act, d.slope, d.Cout = act, slope, out.Cphys
        d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
        self.keep += [d, packed, sc, sh, sw, sb]
        fn = self.lib.cp_conv3x3_halo_seg
        dref = C.byref(d)
        pw, ps, pt, psw, psb = packed.data_ptr(), sc.data_ptr(), sh.data_ptr(), sw.data_ptr(), sb.data_ptr()
        xtb, otb = x.tbuf, out.tbuf
        self._add(fn, lambda P: (dref, P(xtb), pw, ps, pt, P(otb), psw, psb, S, P(seg_tbuf)), "conv3x3_halo4:" + wkey, [xtb], [otb, seg_tbuf])
        fl = 2 * x.B * x.H * x.W * 9 * wCin * wCout
        self.flops += fl
        nbytes = x.B * x.H * x.W * (wCin + wCout) * self.es + 9 * wCin * wCout * self.es
        self.conv_log.append((wkey, x.B * x.H * x.W, wCout, 9 * wCin, fl, "conv3x3_halo4", nbytes))
        return out

    def can_fuse_basicblock(self, x: Act, C_):
        return (USE_FUSED_BB and C_ <= 32 and x.Cphys <= 4 * self.E and x.W >= 16 and x.H >= 8 and x.C == C_)

    def basicblock_fused(self, x: Act, k1, w1, s1, t1, k2, w2, s2, t2):
        """relu(bn2(conv2(relu(bn1(conv1(x))))) + x) in one launch (cp_basicblock_fused)."""
        C_ = w1.shape[0]
        pw1 = self.ws.pack_rows3x3(k1, w1, C_, C_, x.Cphys)
        pw2 = self.ws.pack_halo(k2, w2, C_, C_, x.Cphys)
        a1 = self.ws.affine(k1 + "#0", s1, t1, C_)
        a2 = self.ws.affine(k2 + "#0", s2, t2, C_)
        out = self.act(x.H, x.W, C_)
        d = CpConvDesc()
        d.dtype, d.out_f32, d.B, d.H, d.W = self.dtype, 0, x.B, x.H, x.W
        d.Cin, d.in_cstride, d.in_coff = x.Cphys, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 3, 3, 1, 1, x.H, x.W
        d.Cout, d.act, d.slope = out.Cphys, ACT_RELU, 0.0
        d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
        self.keep += [d, pw1, pw2, a1, a2]
        fn = self.lib.cp_basicblock_fused
        dref = C.byref(d)
        ptrs = (pw1.data_ptr(), a1[0].data_ptr(), a1[1].data_ptr(), pw2.data_ptr(), a2[0].data_ptr(), a2[1].data_ptr())
        xtb, otb = x.tbuf, out.tbuf
        self._add(fn, lambda P: (dref, P(xtb)) + ptrs + (P(otb),), "basicblock_fused:" + k1, [xtb], [otb])
        fl = 2 * x.B * x.H * x.W * 9 * C_ * C_
        nb = x.B * x.H * x.W * C_ * self.es
        self.flops += 2 * fl
        self.conv_log.append((k1, x.B * x.H * x.W, C_, 9 * C_, 2 * fl, "basicblock_fused", 2 * nb + 2 * 9 * C_ * C_ * self.es))
        return out

    def can_fuse_stem(self, size):
        return USE_STEM and self.dtype == CP_BF16 and self.B >= self.stem_min and size % 64 == 0

    def hr_stem(self, img_t, size, k1, w1, s1, t1, k2, w2, s2, t2):
        """timm hrnet stem (conv1-bn1-relu-conv2-bn2-relu) from the NCHW fp32 image in one launch (cp_hr_stem)"""
        ck = ("stem", k1)
        if ck not in self.ws.cache:
            p1 = torch.empty(self.lib.cp_hr_stem_weight_bytes(0), dtype=torch.uint8, device=self.device)
            p2 = torch.empty(self.lib.cp_hr_stem_weight_bytes(1), dtype=torch.uint8, device=self.device)
            w1c, w2c = w1.contiguous(), w2.contiguous()
            self.ws.keep += [w1c, w2c]
            st = torch.cuda.current_stream(self.device).cuda_stream
            _abi.check(self.lib.cp_pack_hr_stem_weights(st, w1c.data_ptr(), w2c.data_ptr(), p1.data_ptr(), p2.data_ptr()), "cp_pack_hr_stem_weights")
            self.ws.cache[ck] = (p1, p2)
        p1, p2 = self.ws.cache[ck]
        a1, a2 = self.ws.affine(k1 + "#0", s1, t1, 64), self.ws.affine(k2 + "#0", s2, t2, 64)
        out = self.act(size // 4, size // 4, 64)
        self.keep += [p1, p2, a1, a2]
        fn = self.lib.cp_hr_stem
        ip = img_t.data_ptr()
        ot = out.tbuf
        args = (p1.data_ptr(), a1[0].data_ptr(), a1[1].data_ptr(), p2.data_ptr(), a2[0].data_ptr(), a2[1].data_ptr())
        self._add(fn, lambda P: (ip, self.B, size, size) + args + (P(ot),), "hr_stem:" + k1, [], [ot])
        fl = 2 * self.B * ((size // 2) ** 2 * 64 * 27 + (size // 4) ** 2 * 64 * 576)
        self.flops += fl
        self.conv_log.append((k1, self.B * (size // 4) ** 2, 64, 576, fl, "hr_stem", self.B * (3 * size * size * 4 + (size // 4) ** 2 * 64 * 2)))
        return out

    def can_chain(self, C_, H, W):
        return (USE_CHAIN and self.dtype == CP_BF16 and self.B >= self.chain_min and bool(self.lib.cp_hr_chain_supported(C_, H, W)))

    def can_chain_tail(self, C_, H, W, convs):
        """the stride-2 fuse convs that read this branch's output fit the chain launch's tail (cp_hr_branch_chain_tail)"""
        return (USE_CHAIN_TAIL and self.can_chain(C_, H, W) and 1 <= len(convs) <= 3 and bool(self.lib.cp_hr_chain_tail_supported(C_, H, W))
                and sum(_rup(c[1].shape[0], self.E) for c in convs) <= self.lib.cp_hr_chain_tail_channels())

    def hr_chain(self, name, srcs, shifts, relu_in, ws, affs, C_, H, W, tail=None):
        """4 BasicBlocks of an HRNet branch on relu(sum of `srcs` (nearest-upsampled by 2^shift)) in ONE launch
        (cp_hr_branch_chain): ws / affs = the 8 fp32 conv weights and folded-BN (scale, shift) pairs in execution order.
        tail: [(wkey, w fp32 (Cout, C, 3, 3), scale, shift, relu)] -- the 3x3 / stride 2 fuse-layer convs reading the branch output, run in
        the same launch off the map in LDS (can_chain_tail); returns (out, [their outputs]) then."""
        blob, aff = self.ws.pack_chain(name, ws, affs, C_, H, W)
        out = self.act(H, W, C_)
        if tail is not None:
            return self._hr_chain_tail(name, srcs, shifts, relu_in, blob, aff, out, C_, H, W, tail)
        n = len(srcs)
        arr_p = (C.c_void_p * 4)()
        arr_s = (C.c_int32 * 4)(*([int(v) for v in shifts] + [0] * (4 - n)))
        self.keep += [arr_p, arr_s, blob, aff]
        for s_, sh in zip(srcs, shifts):
            assert s_.coff == 0 and s_.cstride == s_.Cphys == out.Cphys and (s_.H << sh, s_.W << sh) == (H, W)
        tbs = [s_.tbuf for s_ in srcs]
        ot = out.tbuf
        fn = self.lib.cp_hr_branch_chain
        bp, ap = blob.data_ptr(), aff.data_ptr()

        def argb(P):
            for i, t in enumerate(tbs):
                arr_p[i] = P(t)
            return (self.B, C_, H, W, n, arr_p, arr_s, 1 if relu_in else 0, bp, ap, P(ot))

        self._add(fn, argb, "hr_chain:" + name, tbs, [ot])
        fl = 8 * 2 * self.B * H * W * 9 * C_ * C_
        self.flops += fl
        nb = self.B * H * W * C_ * self.es
        self.conv_log.append((name, self.B * H * W, C_, 9 * C_, fl, "hr_chain", (n + 1) * nb + 8 * 9 * C_ * C_ * self.es))
        return out

    def can_chain_tails(self, C_, H, W, convs):
        """every first-level fuse conv fed by this (36 / 72 / 144-channel) branch fits its chain launch's tail (cp_hr_branch_chain_tails);
        convs: [(wkey, w, scale, shift, k in (1, 3), relu)] as for hr_fuse_out"""
        return (USE_CHAIN_TAILS and self.can_chain(C_, H, W) and 1 <= len(convs) <= 3 and
                all(bool(self.lib.cp_hr_chain_tailconv_supported(C_, H, W, 1 if c[4] == 3 else 0, int(c[1].shape[0]))) for c in convs))

    def hr_chain_tails(self, name, srcs, shifts, relu_in, ws, affs, C_, H, W, convs):
        """hr_chain + the module's first-level fuse convs that read this branch, in the launch's tail off the map in LDS; returns
        (out, [conv outputs]) -- the outputs hr_fuse_out would have produced (same shapes, bf16 rounding of a different K order)"""
        lib = self.lib
        blob, aff = self.ws.pack_chain(name, ws, affs, C_, H, W)
        out = self.act(H, W, C_)
        arr = (CpChainTailConv * len(convs))()
        touts, keep, fl_t, nb_t = [], [], 0, 0
        for i, (wkey, w, scale, shift, k, relu) in enumerate(convs):
            kind = 1 if k == 3 else 0
            Cout = int(w.shape[0])
            ck = ("chain_tailconv", wkey)
            if ck not in self.ws.cache:
                buf = torch.empty(lib.cp_hr_chain_tailconv_weight_bytes(C_, H, W, kind, Cout), dtype=torch.uint8, device=self.device)
                wc = w.contiguous()
                sc = scale.to(device=self.device, dtype=torch.float32).contiguous()
                self.ws.keep += [wc, sc]
                st = torch.cuda.current_stream(self.device).cuda_stream
                _abi.check(lib.cp_pack_hr_chain_tailconv_weight(st, wc.data_ptr(), sc.data_ptr(), C_, H, W, kind, Cout, buf.data_ptr()),
                           "cp_pack_hr_chain_tailconv_weight(%s)" % wkey)
                sh = torch.zeros(_rup(Cout, 16), dtype=torch.float32, device=self.device)
                sh[:Cout] = shift
                self.ws.cache[ck] = (buf, sh)
            buf, sh = self.ws.cache[ck]
            o = self.act(H >> kind, W >> kind, Cout)
            assert o.coff == 0 and o.cstride == o.Cphys
            arr[i].packed_w, arr[i].shift = buf.data_ptr(), sh.data_ptr()
            arr[i].kind, arr[i].Cout, arr[i].out_cphys, arr[i].relu = kind, Cout, o.Cphys, 1 if relu else 0
            touts.append(o)
            keep += [buf, sh]
            fl_t += 2 * self.B * o.H * o.W * k * k * C_ * Cout
            nb_t += self.B * o.H * o.W * Cout * self.es + k * k * C_ * Cout * self.es
        n = len(srcs)
        arr_p = (C.c_void_p * 4)()
        arr_s = (C.c_int32 * 4)(*([int(v) for v in shifts] + [0] * (4 - n)))
        self.keep += [arr_p, arr_s, blob, aff, arr] + keep
        for s_, sh_ in zip(srcs, shifts):
            assert s_.coff == 0 and s_.cstride == s_.Cphys == out.Cphys and (s_.H << sh_, s_.W << sh_) == (H, W)
        tbs, ot, tts = [s_.tbuf for s_ in srcs], out.tbuf, [o.tbuf for o in touts]
        bp, ap = blob.data_ptr(), aff.data_ptr()

        def argb(P):
            for i, t in enumerate(tbs):
                arr_p[i] = P(t)
            for i, t in enumerate(tts):
                arr[i].out = P(t)
            return (self.B, C_, H, W, n, arr_p, arr_s, 1 if relu_in else 0, bp, ap, P(ot), len(convs), arr)

        self._add(lib.cp_hr_branch_chain_tails, argb, "hr_chain:" + name, tbs, [ot] + tts)
        fl = 8 * 2 * self.B * H * W * 9 * C_ * C_ + fl_t
        self.flops += fl
        nb = self.B * H * W * C_ * self.es
        self.conv_log.append((name, self.B * H * W, C_, 9 * C_, fl, "hr_chain", (n + 1) * nb + 8 * 9 * C_ * C_ * self.es + nb_t))
        return out, touts

    def _hr_chain_tail(self, name, srcs, shifts, relu_in, blob, aff, out, C_, H, W, tail):
        lib = self.lib
        ck = ("chain_tail", name)
        if ck not in self.ws.cache:
            tb = torch.zeros(lib.cp_hr_chain_tail_weight_bytes(), dtype=torch.uint8, device=self.device)
            tsh = torch.zeros(lib.cp_hr_chain_tail_channels(), dtype=torch.float32, device=self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            piece = 0
            for wkey, w, scale, shift, relu in tail:
                Cout = w.shape[0]
                cph = _rup(Cout, self.E)
                wc = w.contiguous()
                sc = scale.to(device=self.device, dtype=torch.float32).contiguous()
                self.ws.keep += [wc, sc]
                _abi.check(lib.cp_pack_hr_chain_tail_weight(st, wc.data_ptr(), sc.data_ptr(), Cout, piece, cph, tb.data_ptr()),
                           "cp_pack_hr_chain_tail_weight(%s)" % wkey)
                tsh[piece * 8: piece * 8 + Cout] = shift
                piece += cph // 8
            self.ws.cache[ck] = (tb, tsh)
        tb, tsh = self.ws.cache[ck]
        touts = [self.act(H >> 1, W >> 1, w.shape[0]) for _, w, *_ in tail]
        tl = CpChainTail()
        tl.packed_w, tl.shift, tl.nconv = tb.data_ptr(), tsh.data_ptr(), len(tail)
        for i, ((wkey, w, scale, shift, relu), o) in enumerate(zip(tail, touts)):
            assert o.coff == 0 and o.cstride == o.Cphys
            tl.Cout[i], tl.out_cphys[i], tl.relu[i] = w.shape[0], o.Cphys, 1 if relu else 0
        n = len(srcs)
        arr_p = (C.c_void_p * 4)()
        arr_s = (C.c_int32 * 4)(*([int(v) for v in shifts] + [0] * (4 - n)))
        self.keep += [arr_p, arr_s, blob, aff, tb, tsh, tl]
        for s_, sh in zip(srcs, shifts):
            assert s_.coff == 0 and s_.cstride == s_.Cphys == out.Cphys and (s_.H << sh, s_.W << sh) == (H, W)
        tbs, ot, tts = [s_.tbuf for s_ in srcs], out.tbuf, [o.tbuf for o in touts]
        bp, ap = blob.data_ptr(), aff.data_ptr()

        def argb(P):
            for i, t in enumerate(tbs):
                arr_p[i] = P(t)
            for i, t in enumerate(tts):
                tl.out[i] = P(t)
            return (self.B, C_, H, W, n, arr_p, arr_s, 1 if relu_in else 0, bp, ap, P(ot), C.byref(tl))

        self._add(lib.cp_hr_branch_chain_tail, argb, "hr_chain:" + name, tbs, [ot] + tts)
        fl = 8 * 2 * self.B * H * W * 9 * C_ * C_ + sum(2 * self.B * (H >> 1) * (W >> 1) * 9 * C_ * w.shape[0] for _, w, *_ in tail)
        self.flops += fl
        nb = self.B * H * W * C_ * self.es
        nbt = sum(self.B * (H >> 1) * (W >> 1) * w.shape[0] * self.es + 9 * C_ * w.shape[0] * self.es for _, w, *_ in tail)
        self.conv_log.append((name, self.B * H * W, C_, 9 * C_, fl, "hr_chain", (n + 1) * nb + 8 * 9 * C_ * C_ * self.es + nbt))
        return out, touts

    def can_fuse_out(self, x: Act, couts=()):
        """couts: output channels of the convs the launch would run (each must fit the kernel's epilogue table: <= 160 padded)"""
        return (USE_FUSE_OUT and self.dtype == CP_BF16 and self.B >= self.fuse_out_min and x.coff == 0 and x.cstride == x.Cphys
                and bool(self.lib.cp_hr_fuse_out_supported(x.H, x.W, x.Cphys))
                and all(self.lib.cp_hr_fuse_out_affine_floats(_rup(int(c), self.E)) > 0 for c in couts))

    def hr_fuse_out(self, x: Act, convs):
        """Every first-level fuse-layer conv fed by one branch output `x` in ONE launch (cp_hr_fuse_out).  convs: list of
        (wkey, w fp32 (Cout, Cin, k, k), scale, shift, k in (1, 3), relu): k = 1 -> 1x1 at x's resolution, k = 3 -> 3x3 / stride 2.
        Returns the output Acts."""
        assert 1 <= len(convs) <= 4
        arr = (CpFuseConv * len(convs))()
        outs, keep, fl, nbytes = [], [], 0, x.B * x.H * x.W * x.C * self.es
        for i, (wkey, w, scale, shift, k, relu) in enumerate(convs):
            kind = 1 if k == 3 else 0
            Cout, Cin = w.shape[0], w.shape[1]
            assert Cin == x.C and w.shape[2] == k
            out = self.act(x.H >> kind, x.W >> kind, Cout)
            ck = ("fuse_out", wkey)
            if ck not in self.ws.cache:
                buf = torch.empty(self.lib.cp_hr_fuse_out_weight_bytes(x.Cphys, out.Cphys, kind), dtype=torch.uint8, device=self.device)
                wc = w.contiguous()
                self.ws.keep.append(wc)
                st = torch.cuda.current_stream(self.device).cuda_stream
                _abi.check(self.lib.cp_pack_hr_fuse_out_weight(st, wc.data_ptr(), Cout, Cin, x.Cphys, out.Cphys, kind, buf.data_ptr()),
                           "cp_pack_hr_fuse_out_weight")
                n = self.lib.cp_hr_fuse_out_affine_floats(out.Cphys)
                aff = torch.zeros(2, n, dtype=torch.float32, device=self.device)
                aff[0, :Cout] = scale
                aff[1, :Cout] = shift
                self.ws.cache[ck] = (buf, aff)
            buf, aff = self.ws.cache[ck]
            keep += [buf, aff]
            arr[i].packed_w, arr[i].affine = buf.data_ptr(), aff.data_ptr()
            arr[i].kind, arr[i].Cout, arr[i].out_cphys, arr[i].relu = kind, Cout, out.Cphys, 1 if relu else 0
            outs.append(out)
            fl += 2 * x.B * out.H * out.W * k * k * Cin * Cout
            nbytes += x.B * out.H * out.W * Cout * self.es + k * k * Cin * Cout * self.es
        self.keep += keep + [arr]
        fn = self.lib.cp_hr_fuse_out
        xt, ots = x.tbuf, [o.tbuf for o in outs]

        def argb(P):
            for i, t in enumerate(ots):
                arr[i].out = P(t)
            return (P(xt), x.B, x.H, x.W, x.Cphys, len(convs), arr)

        name = convs[0][0]
        self._add(fn, argb, "hr_fuse_out:" + name, [xt], ots)
        self.flops += fl
        self.conv_log.append((name, x.B * x.H * x.W, sum(c[1].shape[0] for c in convs), x.C, fl, "hr_fuse_out", nbytes))
        return outs

    def can_fuse_bottleneck(self, x: Act, planes, cout, has_ds):
        cin = 64 if has_ds else 256
        return (USE_FUSED_BN and self.dtype == CP_BF16 and x.C == cin and x.Cphys == cin and planes == 64 and cout == 256
                and x.B * x.H * x.W * x.cstride * 2 < (1 << 31))

    def bottleneck_fused(self, x: Act, keys, ws, affs, out: Act = None):
        """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)) in one launch (cp_bottleneck_fused).
        keys/ws/affs: (conv1, conv2, conv3[, downsample]) cache keys, fp32 weights and folded (scale, shift) pairs."""
        has_ds = len(keys) == 4
        cin = x.C
        pw = [self.ws.pack(keys[0], ws[0], 64, cin, 1, 1, cin, 64), self.ws.pack(keys[1], ws[1], 64, 64, 3, 3, 64, 64),
              self.ws.pack(keys[2], ws[2], 256, 64, 1, 1, 64, 256)]
        if has_ds:
            pw.append(self.ws.pack(keys[3], ws[3], 256, 64, 1, 1, 64, 256))
        af = [self.ws.affine(keys[i] + "#0", affs[i][0], affs[i][1], (64, 64, 256, 256)[i]) for i in range(len(keys))]
        if out is None:
            out = self.act(x.H, x.W, 256)
        d = CpConvDesc()
        d.dtype, d.out_f32, d.B, d.H, d.W = self.dtype, 0, x.B, x.H, x.W
        d.Cin, d.in_cstride, d.in_coff = cin, x.cstride, x.coff
        d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 3, 3, 1, 1, x.H, x.W
        d.Cout, d.act, d.slope = 256, ACT_RELU, 0.0
        d.o_base, d.o_sb, d.o_sy, d.o_sx, d.o_sc = out.coff, out.H * out.W * out.cstride, out.W * out.cstride, out.cstride, 1
        self.keep += [d] + pw + af
        fn = self.lib.cp_bottleneck_fused
        dref = C.byref(d)
        ptrs = ()
        for i in range(len(keys)):
            ptrs += (pw[i].data_ptr(), af[i][0].data_ptr(), af[i][1].data_ptr())
        if not has_ds:
            ptrs += (None, None, None)
        xtb, otb = x.tbuf, out.tbuf
        self._add(fn, lambda P: (dref, P(xtb)) + ptrs + (P(otb),), "bottleneck_fused:" + keys[0], [xtb], [otb])
        npx = x.B * x.H * x.W
        macs = cin * 64 + 9 * 64 * 64 + 64 * 256 + (64 * 256 if has_ds else 0)
        fl = 2 * npx * macs
        self.flops += fl
        self.conv_log.append((keys[0], npx, 256, cin, fl, "bottleneck_fused", npx * (cin + 256) * self.es + macs * self.es))
        return out

    def upsample2x(self, x: Act, out: Act):
        fn = self.lib.cp_upsample2x_bilinear_ac
        a = (self.dtype,)
        xt, ot = x.tbuf, out.tbuf
        tail = (x.B, x.H, x.W, x.Cphys, x.cstride, x.coff, out.cstride, out.coff)
        self._add(fn, lambda P: a + (P(xt), P(ot)) + tail, "upsample2x", [xt], [ot])
        return out

    def fuse_sum(self, srcs, shifts, out: Act, relu=True):
        n = len(srcs)
        arr_p = (C.c_void_p * 4)()
        arr_s = (C.c_int32 * 4)(*([int(s) for s in shifts] + [0] * (4 - n)))
        self.keep += [arr_p, arr_s]
        fn = self.lib.cp_fuse_sum_act
        tbs = [s.tbuf for s in srcs]
        for s in srcs:
            assert s.coff == 0 and s.cstride == s.Cphys == out.Cphys
        ot = out.tbuf

        def argb(P):
            for i, t in enumerate(tbs):
                arr_p[i] = P(t)
            return (self.dtype, n, arr_p, arr_s, P(ot), out.B, out.H, out.W, out.Cphys, 1 if relu else 0, out.cstride, out.coff)

        self._add(fn, argb, "fuse_sum", tbs, [ot])
        return out

    def maxpool(self, x: Act):
        out = self.act(x.H // 2, x.W // 2, x.C)
        fn = self.lib.cp_maxpool3x3s2
        xt, ot = x.tbuf, out.tbuf
        self._add(fn, lambda P: (self.dtype, P(xt), P(ot), x.B, x.H, x.W, x.Cphys), "maxpool", [xt], [ot])
        return out

    def edge_gather(self, pq: Act, idx_t, gids_t, out: Act, K, C_, G, slope):
        gdt = self.gdt
        fn = self.lib.cp_edgeconv_gather_max
        pt, ot = pq.tbuf, out.tbuf
        ip = idx_t.data_ptr()
        gp = gids_t.data_ptr() if gids_t is not None else None
        N = pq.W
        self._add(fn, lambda P: (gdt, P(pt), ip, gp, P(ot), pq.B, N, K, C_, G, out.cstride, out.coff, slope),
                  "edge_gather:%d" % C_, [pt], [ot])
        return out

    def can_fuse_edgeconv(self, N, K, Cin, Cout):
        return (USE_EDGE_FUSED and self.dtype == CP_BF16 and self.B >= self.edge_min
                and bool(self.lib.cp_edgeconv_fused_supported(N, K, Cin, Cout)))

    def edge_fused(self, x: Act, wkey, wpq, scale, shift, idx_t, gids_t, out: Act, K, G, slope):
        """whole EdgeConv layer in one launch (cp_edgeconv_fused): wpq fp32 (2 C', C) = [W1 ; W2 - W1]"""
        Co2, Cin = wpq.shape[0], wpq.shape[1]
        gdt = self.gdt
        ck = ("edge_fused", wkey, gdt)
        if ck not in self.ws.cache:
            buf = torch.empty(self.lib.cp_edgeconv_fused_weight_bytes(Cin, Co2 // 2), dtype=torch.uint8, device=self.device)
            w2 = wpq.reshape(Co2, Cin).contiguous()
            self.ws.keep.append(w2)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _abi.check(self.lib.cp_pack_edgeconv_fused_weight_t(st, gdt, w2.data_ptr(), Cin, Co2 // 2, buf.data_ptr()), "cp_pack_edgeconv_fused_weight")
            self.ws.cache[ck] = (buf, scale.contiguous(), shift.contiguous())
        buf, sc, sh = self.ws.cache[ck]
        self.keep += [buf, sc, sh]
        fn = self.lib.cp_edgeconv_fused_t
        xt, ot = x.tbuf, out.tbuf
        sk = ("edge_sched", idx_t.data_ptr(), tuple(idx_t.shape))
        if EDGE_SCHED and sk not in self.ws.cache:      # the same graph, every keypoint's list reordered against LDS bank conflicts
            from .graph_sched import schedule_neighbours
            i3 = idx_t.detach().cpu().numpy().reshape(-1, x.W, K)
            self.ws.cache[sk] = torch.from_numpy(schedule_neighbours(i3)[0]).reshape(tuple(idx_t.shape)).to(self.device).contiguous()
        if EDGE_SCHED:
            idx_t = self.ws.cache[sk]
            self.keep.append(idx_t)
        ip = idx_t.data_ptr()
        gp = gids_t.data_ptr() if gids_t is not None else None
        N = x.W
        a = (buf.data_ptr(), sc.data_ptr(), sh.data_ptr(), ip, gp)
        self._add(fn, lambda P: (gdt, P(xt), x.cstride, x.coff) + a + (P(ot), out.cstride, out.coff, self.B, N, K, Cin, Co2 // 2, G, slope),
                  "edge_fused:" + wkey, [xt], [ot])
        fl = 2 * self.B * N * Cin * Co2
        self.flops += fl
        self.conv_log.append((wkey, self.B * N, Co2, Cin, fl, "edge_fused", self.B * N * (Cin + Co2 // 2) * self.es + Co2 * Cin * self.es))
        return out

    def can_tile_edgeconv(self, N, K, Cin, Cout, HPAD):
        """large graphs: one workgroup per (crop, patch of 512 keypoints) -- needs B * N / 512 workgroups to fill the chip"""
        return (USE_EDGE_TILED and self.dtype == CP_BF16 and N > 512 and N % 512 == 0 and self.B * (N // 512) >= self.edge_min
                and bool(self.lib.cp_edgeconv_tiled_supported(N, K, Cin, Cout, HPAD)))

    def edge_tiled(self, x: Act, wkey, wpq, scale, shift, tiled, gids_t, out: Act, K, G, slope):
        """whole EdgeConv layer at N > 512 (cp_edgeconv_tiled: key-table launch + LDS-staged gather launch); x / out rows are in
        the INTERNAL keypoint order of `tiled` (graph_sched.tile_schedule, device tensors)"""
        Co2, Cin = wpq.shape[0], wpq.shape[1]
        Co, N = Co2 // 2, x.W
        st = torch.cuda.current_stream(self.device).cuda_stream
        gdt = self.gdt
        ck = ("edge_fused", wkey, gdt)
        if ck not in self.ws.cache:
            buf = torch.empty(self.lib.cp_edgeconv_fused_weight_bytes(Cin, Co), dtype=torch.uint8, device=self.device)
            w2 = wpq.reshape(Co2, Cin).contiguous()
            self.ws.keep.append(w2)
            _abi.check(self.lib.cp_pack_edgeconv_fused_weight_t(st, gdt, w2.data_ptr(), Cin, Co, buf.data_ptr()), "cp_pack_edgeconv_fused_weight")
            self.ws.cache[ck] = (buf, scale.contiguous(), shift.contiguous())
        buf, sc, sh = self.ws.cache[ck]
        cq = ("edge_tiled_q", wkey, gdt)
        if cq not in self.ws.cache:
            bq = torch.empty(self.lib.cp_edgeconv_tiled_weight_bytes(Cin, Co), dtype=torch.uint8, device=self.device)
            w2 = wpq.reshape(Co2, Cin).contiguous()
            self.ws.keep.append(w2)
            _abi.check(self.lib.cp_pack_edgeconv_tiled_weight_t(st, gdt, w2.data_ptr(), Cin, Co, bq.data_ptr()), "cp_pack_edgeconv_tiled_weight")
            self.ws.cache[cq] = bq
        bq = self.ws.cache[cq]
        ktab = self.tensor(self.lib.cp_edgeconv_tiled_table_bytes(self.B, N, Co), es=1)
        self.keep += [buf, sc, sh, bq, tiled["halo"], tiled["nbr"]]
        fn = self.lib.cp_edgeconv_tiled_t
        xt, ot = x.tbuf, out.tbuf
        gp = gids_t.data_ptr() if gids_t is not None else None
        a = (buf.data_ptr(), bq.data_ptr(), sc.data_ptr(), sh.data_ptr(), tiled["halo"].data_ptr(), tiled["nbr"].data_ptr(), gp)
        HPAD = tiled["HPAD"]
        self._add(fn, lambda P: (gdt, P(xt), x.cstride, x.coff) + a + (P(ktab), P(ot), out.cstride, out.coff, self.B, N, K, Cin, Co, G, HPAD, slope),
                  "edge_tiled:" + wkey, [xt], [ot, ktab])
        fl = 2 * self.B * N * Cin * Co2
        self.flops += fl
        self.conv_log.append((wkey, self.B * N, Co2, Cin, fl, "edge_tiled", self.B * N * (Cin + Co) * self.es + Co2 * Cin * self.es))
        return out

    def can_fuse_query_mlp(self, x: Act, dims):
        return (USE_MLP_FUSED and self.dtype == CP_BF16 and x.H == 1 and self.B * x.W >= self.mlp_min_rows
                and len(dims) == 4 and bool(self.lib.cp_mlp_query_fused_supported(*[int(d) for d in dims])) and x.C == dims[0])

    def mlp_query_fused(self, x: Act, keys, ws, bs, slope, out_tbuf, ostr):
        """MLP_QueryNet (pipeline.py:168-180) in one launch: ws = the three nn.Linear weights, bs their biases; the two logits of
        row (b, n) go to out_tbuf[ostr[0] + b * ostr[1] + n * ostr[3] + c * ostr[4]] (fp32)"""
        N = x.W
        gdt = self.gdt
        pw1 = self.ws.pack_gemm(keys[0], ws[0].reshape(ws[0].shape[0], ws[0].shape[1], 1, 1), 256, 256, 256, dtype=gdt)
        pw2 = self.ws.pack_gemm(keys[1], ws[1].reshape(ws[1].shape[0], ws[1].shape[1], 1, 1), 64, 256, 256, dtype=gdt)
        ck = ("mlp_query", keys[0])
        if ck not in self.ws.cache:
            dev = self.device
            self.ws.cache[ck] = (torch.ones(256, dtype=torch.float32, device=dev), bs[0].float().contiguous(),
                                 torch.ones(64, dtype=torch.float32, device=dev), bs[1].float().contiguous(),
                                 ws[2].float().reshape(2, 64).contiguous(), bs[2].float().contiguous())
        s1, t1, s2, t2, w3, b3 = self.ws.cache[ck]
        self.keep += [pw1, pw2, s1, t1, s2, t2, w3, b3]
        xt = x.tbuf
        a1 = (pw1.data_ptr(), s1.data_ptr(), t1.data_ptr(), float(slope), pw2.data_ptr(), s2.data_ptr(), t2.data_ptr(), float(slope),
              w3.data_ptr(), b3.data_ptr())
        o = (int(ostr[0]), int(ostr[1]), int(ostr[3]), int(ostr[4]))
        self._add(self.lib.cp_mlp_query_fused_t, lambda P: (gdt, P(xt), x.cstride, x.coff, self.B, N) + a1 + (P(out_tbuf),) + o,
                  "mlp_fused:" + keys[0], [xt], [out_tbuf])
        M = self.B * N
        fl = 2 * M * (256 * 256 + 256 * 64 + 64 * 2)
        self.flops += fl
        self.conv_log.append((keys[0], M, 256 + 64 + 2, 256, fl, "mlp_fused", M * (256 * self.es + 8) + (256 * 256 + 64 * 256) * self.es))

    def can_fuse_mlp_pair(self, x: Act, w1, w2):
        return (USE_MLP_FUSED and self.dtype == CP_BF16 and x.H == 1 and self.B * x.W >= self.mlp_min_rows and x.C == w1.shape[1]
                and x.C == x.Cphys and bool(self.lib.cp_mlp_pair_fused_supported(int(w1.shape[1]), int(w1.shape[0]), int(w2.shape[0])))
                and w2.shape[1] == w1.shape[0])

    def mlp_pair_fused(self, x: Act, keys, ws, bs, slope, out: Act = None):
        """pre_graph_module (two nn.Linear + LeakyReLU) in one launch: the 256-channel hidden rows stay on chip"""
        N, Cin = x.W, x.C
        gdt = self.gdt
        pw1 = self.ws.pack_gemm(keys[0], ws[0].reshape(256, Cin, 1, 1), 256, Cin, Cin, dtype=gdt)
        pw2 = self.ws.pack_gemm(keys[1], ws[1].reshape(256, 256, 1, 1), 256, 256, 256, dtype=gdt)
        ck = ("mlp_pair", keys[0])
        if ck not in self.ws.cache:
            self.ws.cache[ck] = (bs[0].float().contiguous(), bs[1].float().contiguous())
        b1, b2 = self.ws.cache[ck]
        self.keep += [pw1, pw2, b1, b2]
        if out is None:
            out = self.act(1, N, 256)
        xt, ot = x.tbuf, out.tbuf
        a1 = (pw1.data_ptr(), b1.data_ptr(), float(slope), pw2.data_ptr(), b2.data_ptr(), float(slope))
        self._add(self.lib.cp_mlp_pair_fused_t, lambda P: (gdt, P(xt), x.cstride, x.coff, Cin, self.B, N) + a1 + (P(ot), out.cstride, out.coff),
                  "mlp_fused:" + keys[0], [xt], [ot])
        M = self.B * N
        fl = 2 * M * (Cin * 256 + 256 * 256)
        self.flops += fl
        self.conv_log.append((keys[0], M, 512, Cin, fl, "mlp_fused", M * (Cin + 256) * self.es + (Cin * 256 + 256 * 256) * self.es))
        return out

    def can_fuse_mlp_pair_gather(self, L: Act, patches: Act, w1, w2, E_ch, k):
        """the pair with Index2Feat's gather in its loader (cp_mlp_pair_fused_gather): L = the [local 4 x 64 | graph] concat buffer"""
        return (USE_MLP_GATHER and self.can_fuse_mlp_pair(L, w1, w2) and patches is not None and E_ch == 64 and L.C > 256 and L.W >= 4 and (L.W & (L.W - 1)) == 0
                and bool(self.lib.cp_mlp_pair_fused_gather_supported(int(L.C - 256), int(E_ch), int(k))))

    def mlp_pair_fused_gather(self, patches: Act, xid_t, yid_t, mask_t, L: Act, keys, ws, bs, slope, N, k, out: Act = None):
        """Index2Feat_module's 4-tap gather x RoI bit (pipeline.py:156-163,280) + pre_graph_module in one launch: the loader of the
        fused pair fetches the taps straight from patch_generator's map, the first 256 channels of `L` are never written"""
        Cin = L.C
        Cg = Cin - 256
        gdt = self.gdt
        pw1 = self.ws.pack_gemm(keys[0], ws[0].reshape(256, Cin, 1, 1), 256, Cin, Cin, dtype=gdt)
        pw2 = self.ws.pack_gemm(keys[1], ws[1].reshape(256, 256, 1, 1), 256, 256, 256, dtype=gdt)
        ck = ("mlp_pair", keys[0])
        if ck not in self.ws.cache:
            self.ws.cache[ck] = (bs[0].float().contiguous(), bs[1].float().contiguous())
        b1, b2 = self.ws.cache[ck]
        zk = ("zeros128",)
        if zk not in self.ws.cache:
            self.ws.cache[zk] = torch.zeros(256, dtype=torch.uint8, device=self.device)
        zeros = self.ws.cache[zk]
        self.keep += [pw1, pw2, b1, b2, zeros]
        if out is None:
            out = self.act(1, N, 256)
        pt, lt, ot = patches.tbuf, L.tbuf, out.tbuf
        a1 = (pw1.data_ptr(), b1.data_ptr(), float(slope), pw2.data_ptr(), b2.data_ptr(), float(slope))
        descs = []

        def args(P):
            g = _abi.CpI2fGather()
            g.patches, g.x_id, g.y_id, g.mask, g.zeros = P(pt), xid_t.data_ptr(), yid_t.data_ptr(), mask_t.data_ptr(), zeros.data_ptr()
            g.p_cstride, g.p_coff, g.Hp, g.Wp, g.k = patches.cstride, patches.coff, patches.H, patches.W, k
            descs.append(g)                                   # the descriptor must outlive the (possibly deferred) call
            return (gdt, C.byref(g), P(lt), L.cstride, L.coff + 256, Cg, self.B, N) + a1 + (P(ot), out.cstride, out.coff)
        self.keep.append(descs)
        self._add(self.lib.cp_mlp_pair_fused_gather_t, args, "mlp_fused:" + keys[0],
                  [pt, lt, self.raw(xid_t), self.raw(yid_t), self.raw(mask_t)], [ot])
        M = self.B * N
        fl = 2 * M * (Cin * 256 + 256 * 256)
        self.flops += fl
        # algorithmic HBM bytes: the gathered taps overlap and repeat, so at most the patch map itself is read once
        nby = (min(patches.B * patches.H * patches.W * 64, M * 256) + M * (Cg + 256) + Cin * 256 + 256 * 256) * self.es
        self.conv_log.append((keys[0], M, 512, Cin, fl, "mlp_fused", nby))
        return out

    def permute_rows(self, x: Act, out: Act, perm_t, gids_t):
        """out[b, i, :] = x[b, perm[g_b, i], :] over whole (B, N, cstride) rows (cp_permute_rows)"""
        assert x.coff == 0 and out.coff == 0 and x.cstride == out.cstride and (x.cstride * self.es) % 16 == 0
        fn = self.lib.cp_permute_rows
        xt, ot = x.tbuf, out.tbuf
        gp = gids_t.data_ptr() if gids_t is not None else None
        self.keep.append(perm_t)
        self._add(fn, lambda P: (P(xt), P(ot), perm_t.data_ptr(), gp, x.B, x.W, x.cstride * self.es), "permute_rows", [xt], [ot])
        return out

    def permute_cols(self, in_t, out_t, perm_t, gids_t, R, N, scatter):
        """(B, R, N) caller-owned arrays: scatter: out[b, r, perm[i]] = in[b, r, i]; else out[b, r, i] = in[b, r, perm[i]]"""
        fn = self.lib.cp_permute_cols
        gp = gids_t.data_ptr() if gids_t is not None else None
        self.keep += [perm_t, in_t, out_t]
        a = (in_t.data_ptr(), out_t.data_ptr(), perm_t.data_ptr(), gp, self.B, R, N, in_t.element_size(), 1 if scatter else 0)
        self._add(fn, lambda P: a, "permute_cols", [self.raw(in_t)], [self.raw(out_t)])

    def can_gather_patch(self, f: Act, N, E_ch, k):
        """gathered patch conv: 4 N rows per crop instead of (H+1)(W+1) positions -- pays from 2 x 4 N < (H+1)(W+1)"""
        return (USE_PATCH_GATHER and self.dtype == CP_BF16 and bool(self.lib.cp_index2feat_conv_supported(f.C, E_ch, k))
                and 8 * N <= (f.H + 1) * (f.W + 1) and f.B * f.H * f.W * f.cstride * 2 < (1 << 31))

    def index2feat_conv(self, f: Act, wkey, w, bias, xid_t, yid_t, mask_t, out: Act, N, k):
        ck = ("patch_gather", wkey)
        if ck not in self.ws.cache:
            buf = torch.empty(self.lib.cp_index2feat_conv_weight_bytes(), dtype=torch.uint8, device=self.device)
            wc = w.contiguous()
            self.ws.keep.append(wc)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _abi.check(self.lib.cp_pack_index2feat_conv_weight(st, wc.data_ptr(), buf.data_ptr()), "cp_pack_index2feat_conv_weight")
            self.ws.cache[ck] = (buf, bias.contiguous())
        buf, bs = self.ws.cache[ck]
        self.keep += [buf, bs]
        fn = self.lib.cp_index2feat_conv_t
        gdt = self.gdt
        ft, ot = f.tbuf, out.tbuf
        a = (buf.data_ptr(), bs.data_ptr(), xid_t.data_ptr(), yid_t.data_ptr(), mask_t.data_ptr())
        self._add(fn, lambda P: (gdt, P(ft), f.cstride, f.coff) + a + (P(ot), f.B, N, f.H, f.W, k, out.cstride, out.coff),
                  "patch_gather:" + wkey, [ft, self.raw(xid_t), self.raw(yid_t), self.raw(mask_t)], [ot])
        fl = 2 * f.B * 4 * N * 64 * 4 * f.C
        self.flops += fl
        # algorithmic HBM bytes: the gathered 2x2 patches overlap and repeat, so at most the feature map itself is read once
        # (PMC: far less -- only pixels some keypoint points at are touched), + the (B, N, 4*64) output + weights
        nby = (f.B * min(f.H * f.W * f.C, 4 * N * 4 * f.C) + f.B * N * 4 * 64 + 64 * 4 * f.C) * self.es
        self.conv_log.append((wkey, f.B * 4 * N, 64, 4 * f.C, fl, "patch_gather", nby))
        return out

    def index2feat(self, patches: Act, xid_t, yid_t, mask_t, out: Act, N, E_ch, k):
        gdt = self.gdt
        fn = self.lib.cp_index2feat_gather
        pt, ot = patches.tbuf, out.tbuf
        args = (xid_t.data_ptr(), yid_t.data_ptr(), mask_t.data_ptr())
        tail = (patches.B, N, patches.H, patches.W, E_ch, k, out.cstride, out.coff)
        self._add(fn, lambda P: (gdt, P(pt)) + args + (P(ot),) + tail, "index2feat", [pt, self.raw(xid_t), self.raw(yid_t), self.raw(mask_t)], [ot])
        return out

    def decode(self, bits_t, stage, mask_t, xid_t, yid_t, x64_t, y64_t, N):
        fn = self.lib.cp_bits_decode
        a = (bits_t.data_ptr(), stage, mask_t.data_ptr(), xid_t.data_ptr(), yid_t.data_ptr(), x64_t.data_ptr(),
             y64_t.data_ptr(), self.B, N)
        self._add(fn, lambda P: a, "decode", [self.raw(bits_t), self.raw(mask_t), self.raw(xid_t), self.raw(yid_t)],
                  [self.raw(mask_t), self.raw(xid_t), self.raw(yid_t), self.raw(x64_t), self.raw(y64_t)])

    def nchw_to_nhwc(self, img_t, Cc, H, W):
        out = self.act(H, W, Cc)
        fn = self.lib.cp_nchw_to_nhwc
        ot = out.tbuf
        ip = img_t.data_ptr()
        self._add(fn, lambda P: (self.dtype, ip, P(ot), self.B, Cc, H, W, out.Cphys), "nchw_to_nhwc", [], [ot])
        return out

    def u8_to_nhwc_norm(self, img_u8_t, H, W, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """uint8 (B,H,W,3) crop -> ToTensor + Normalize -> channels-last activation (cp_u8hwc_to_nhwc_norm)."""
        out = self.act(H, W, 3)
        fn = self.lib.cp_u8hwc_to_nhwc_norm
        ot = out.tbuf
        ip = img_u8_t.data_ptr()
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        self.keep += [m3, s3]
        self._add(fn, lambda P: (self.dtype, ip, P(ot), self.B, H, W, out.Cphys, m3, s3), "u8_to_nhwc_norm", [], [ot])
        return out

    def to_nchw_f32(self, x: Act, out_t):
        fn = self.lib.cp_nhwc_to_nchw_f32
        xt = x.tbuf
        op = out_t.data_ptr()
        self._add(fn, lambda P: (self.dtype, P(xt), op, x.B, x.C, x.H, x.W, x.cstride, x.coff), "to_nchw", [xt], [self.raw(out_t)])

    # ---- memory plan + argument binding
    def _drop_dead_launches(self):
        """A launch whose outputs nobody reads is not run: walking the list backwards, a launch is live if it writes a caller-
        visible (fixed) tensor, declares no outputs (decode / export ops working on raw pointers), or writes a workspace tensor a
        later live launch reads.  In the reference the highest-resolution `incre_modules[0]` feature is computed by timm and
        never used by PoseNet_GNNskip (pipeline.py:354-358 reads img_feats[-1..-3]); with return_img_feats it stays live."""
        needed, dead = set(), set()
        for i in range(len(self.ops) - 1, -1, -1):
            if i not in self._rw:
                continue
            reads, writes = self._rw[i]
            if (not writes) or any(w.fixed is not None or id(w) in needed for w in writes):
                needed.update(id(r) for r in reads)
            else:
                dead.add(i)
        if not dead:
            return 0
        for i in dead:
            self.ops[i] = ("__dead__",)
        for t in self.tbufs:
            t.first = t.last = None
        for i, (reads, writes) in sorted(self._rw.items()):
            if i in dead:
                continue
            for t in reads + writes:
                if t.fixed is None:
                    if t.first is None:
                        t.first = i
                    t.last = i
        live = [(e, k) for e, k in zip(list(self.conv_log), self.conv_log.op_index) if k not in dead]
        self.conv_log[:] = [e for e, _ in live]
        self.conv_log.op_index = [k for _, k in live]
        self.flops = sum(e[4] for e in self.conv_log)
        return len(dead)

    def finalize(self, dce=False):
        """Linear-scan placement of every workspace tensor by liveness (outputs are placed before the op's dead
        inputs are released, so an op never aliases its own operands)."""
        assert self._open is None
        self.dropped = self._drop_dead_launches() if dce else 0
        for t in self.tbufs:                      # inside a fork/join region nothing may be recycled: another lane
            if t.fixed is None and t.first is not None:   # can still be reading (or not yet have written) it
                for rs, re in self.regions:
                    if t.first <= re and t.last >= rs:
                        t.last = max(t.last, re)
        births, deaths = {}, {}
        for t in self.tbufs:
            if t.fixed is None and t.first is not None:
                births.setdefault(t.first, []).append(t)
                deaths.setdefault(t.last, []).append(t)
        free, top = [], 0            # free: list of (offset, size)
        for i in range(len(self.ops)):
            for t in births.get(i, []):
                best = None
                for k, (off, sz) in enumerate(free):
                    if sz >= t.nbytes and (best is None or sz < free[best][1]):
                        best = k
                if best is None:
                    t.offset = top
                    top += t.nbytes
                else:
                    off, sz = free.pop(best)
                    t.offset = off
                    if sz > t.nbytes:
                        free.append((off + t.nbytes, sz - t.nbytes))
            for t in deaths.get(i, []):
                if NO_RECYCLE:
                    continue
                free.append((t.offset, t.nbytes))
                free.sort()
                merged = []
                for off, sz in free:     # coalesce neighbours
                    if merged and merged[-1][0] + merged[-1][1] == off:
                        merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                    else:
                        merged.append((off, sz))
                free = merged
        self.workspace_bytes = top
        self.workspace = torch.empty(max(top, 256), dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()

        def P(t):
            if t.fixed is not None:
                return t.fixed.data_ptr()
            return base + t.offset

        self.calls = []       # kernel launches only: (fn, (None, args...), name)
        self.sched = []       # launches + fork/sync/join markers: ("op", call_index, lane) | ("fork", n) | ...
        acc = []              # per launch: (reads, writes) as (lo, hi, tbuf id) byte intervals -- hazards of run_dag
        for oi, op in enumerate(self.ops):
            if op[0] not in ("__fork__", "__sync__", "__join__", "__mark__", "__wait__", "__dead__"):
                rd, wr = self._rw.get(oi, ([], []))
                acc.append(tuple([(P(t), P(t) + t.nbytes, id(t)) for t in lst] for lst in (rd, wr)))
        self._acc, self._dag = acc, None      # hazards are derived on first use (dag property): quadratic in the launch count
        for op in self.ops:
            if op[0] == "__fork__":
                self.sched.append(("fork", op[1]))
            elif op[0] == "__sync__":
                self.sched.append(("sync", op[1], op[2]))
            elif op[0] == "__join__":
                self.sched.append(("join",))
            elif op[0] == "__mark__":
                self.sched.append(("mark", op[1], op[2]))
            elif op[0] == "__wait__":
                self.sched.append(("wait", op[1], op[2]))
            elif op[0] == "__dead__":
                continue
            else:
                fn, argb, name, lane = op
                self.sched.append(("op", len(self.calls), lane))
                self.calls.append((fn, (None,) + tuple(argb(P)), name))
        return self

    @property
    def dag(self):
        if self._dag is None:
            self._dag = self._hazards(self._acc)
        return self._dag

    @staticmethod
    def _hazards(acc):
        """Per launch, the earlier launches it must wait for (transitively reduced): read-after-write, write-after-read and
        write-after-write on overlapping BYTES (the planner recycles workspace bytes, so two tensors can share them) --
        except that two launches writing the SAME tensor do not order each other: they fill disjoint channel slices of a
        concatenation buffer (an in-place update lists the tensor in its reads and is ordered by that)."""
        def hit(a, b, same_ok):
            for lo, hi, ia in a:
                for lo2, hi2, ib in b:
                    if lo < hi2 and lo2 < hi and not (same_ok and ia == ib):
                        return True
            return False
        deps, anc = [], []
        for k, (rd, wr) in enumerate(acc):
            d = set()
            for j in range(k - 1, -1, -1):
                rj, wj = acc[j]
                if hit(wj, rd, False) or hit(rj, wr, False) or hit(wj, wr, True):
                    d.add(j)
            a = set(d)
            for j in d:
                a |= anc[j]
            anc.append(a)
            deps.append(sorted(j for j in d if not any(j in anc[i] for i in d if i != j)))
        return deps

    def run_dag(self, stream_ptr):
        """Replay on ONE stream that is being captured into a hipGraph, each launch depending on exactly the launches
        whose bytes it reads or overwrites (self.dag) -- the graph is the dataflow DAG of the program: a branch of the next
        HRNet module starts as soon as ITS fuse terms are there, the tail of the slow stride-2 fuse chains runs under it."""
        lib = self.lib
        tails = []                                   # per launch: its last graph node
        cap = 8
        buf, n = (C.c_void_p * cap)(), C.c_int(0)
        used = set()
        for k, (fn, args, name) in enumerate(self.calls):
            nodes = [tails[j] for j in self.dag[k]]
            used.update(self.dag[k])
            arr = (C.c_void_p * max(len(nodes), 1))(*nodes)
            _abi.check(lib.cp_graph_capture_set_deps(stream_ptr, arr, len(nodes)), "capture deps")
            rc = fn(stream_ptr, *args[1:])
            if rc != 0:
                _abi.check(rc, name)
            _abi.check(lib.cp_graph_capture_tail(stream_ptr, buf, cap, C.byref(n)), "capture tail")
            if n.value != 1:
                raise RuntimeError("checkerpose_amd: launch %s left %d capture tail nodes (expected 1)" % (name, n.value))
            tails.append(buf[0])
        sinks = [tails[k] for k in range(len(tails)) if k not in used]
        arr = (C.c_void_p * max(len(sinks), 1))(*sinks)
        _abi.check(lib.cp_graph_capture_set_deps(stream_ptr, arr, len(sinks)), "capture deps")   # join: the capture ends on every sink

    def run(self, stream_ptr):
        """Sequential replay on one stream (program order)."""
        for fn, args, name in self.calls:
            rc = fn(stream_ptr, *args[1:])
            if rc != 0:
                _abi.check(rc, name)

    def run_lanes(self, streams):
        """Replay with fork/join across `streams` (torch.cuda.Stream list, streams[0] = main).  Meant to be called
        while streams[0] is being captured into a hipGraph: the event edges become graph dependencies."""
        ptr = [s.cuda_stream for s in streams]
        active = 1
        events = []          # returned: the caller keeps them alive until the capture has ended
        marks = {}

        def new_event():
            events.append(torch.cuda.Event())
            return events[-1]

        for item in self.sched:
            kind = item[0]
            if kind == "op":
                fn, args, name = self.calls[item[1]]
                rc = fn(ptr[item[2]], *args[1:])
                if rc != 0:
                    _abi.check(rc, name)
            elif kind == "fork":
                active = item[1]
                ev = new_event()
                ev.record(streams[0])
                for k in range(1, active):
                    streams[k].wait_event(ev)
            elif kind == "sync":
                ev = new_event()
                ev.record(streams[item[1]])
                streams[item[2]].wait_event(ev)
            elif kind == "mark":
                marks[item[1]] = new_event()
                marks[item[1]].record(streams[item[2]])
            elif kind == "wait":
                streams[item[2]].wait_event(marks[item[1]])
            else:   # join
                for k in range(1, active):
                    ev = new_event()
                    ev.record(streams[k])
                    streams[0].wait_event(ev)
                active = 1
        return events


class ProgramGroup:
    """Several independent Programs (batch slices of one forward) replayed together.  Sequential replay runs them one
    after the other; under hipGraph capture each gets its own set of lane streams forked from / joined to the main
    stream, so the memory-bound phases of one slice (stem, HRNet body) overlap the MFMA-bound phases of another
    (decoder convs).  Each Program owns its workspace, so slices never alias."""

    def __init__(self, progs):
        self.progs = progs
        self.nlanes = max(p.nlanes for p in progs)

    @property
    def calls(self):
        return [c for p in self.progs for c in p.calls]

    @property
    def conv_log(self):
        return [c for p in self.progs for c in p.conv_log]

    @property
    def flops(self):
        return sum(p.flops for p in self.progs)

    @property
    def workspace_bytes(self):
        return sum(p.workspace_bytes for p in self.progs)

    def run(self, stream_ptr):
        for p in self.progs:
            p.run(stream_ptr)

