"""Runtime glue shared by the drop-in modules: builds / caches the launch Program for a batch size, owns the
persistent I/O tensors, captures the hipGraph and replays it.  Fails loudly -- no CPU or PyTorch fallback.
"""
import ctypes as C
import operator
import os
import weakref

import torch

from .. import _abi
from ..engine import DTYPES, Program, ProgramGroup, WeightStore
from ..netbuilder import NetEmitter, emit_init_net, emit_posenet


_VERSION_OF = operator.attrgetter("_version")
# A parameter / buffer / submodule (re)registration inside a drop-in model's module tree moves THAT model's epoch: its cached
# tensor list of the eval staleness check is rebuilt on the next forward (and the programs dropped if the tensors are not the same
# objects any more) -- a reassigned submodule or parameter must not leave stale tensors in the list.  torch offers the
# registration hooks only process-wide, so the hook itself is global, but it is scoped: a weak map module -> root model, filled
# when a model builds its tensor list; registrations on modules outside every tracked tree (another model, an optimizer's
# containers, torch internals) cost one dictionary miss and touch nothing.  The hooks are installed with the first drop-in model
# and removed again when the last tracked tree is gone.
_TRACKED = weakref.WeakKeyDictionary()          # nn.Module (any node of a tracked tree) -> WeakSet of the root models it sits in
_HOOKS = []


def _bump_epoch(module, *_a):
    roots = _TRACKED.get(module)
    for root in (list(roots) if roots is not None else ()):       # a submodule shared by several drop-in models moves ALL of them
        root._struct_epoch += 1


def _install_hooks():
    if not _HOOKS:
        m = torch.nn.modules.module
        _HOOKS.extend([m.register_module_parameter_registration_hook(_bump_epoch),
                       m.register_module_buffer_registration_hook(_bump_epoch),
                       m.register_module_module_registration_hook(_bump_epoch)])


def _remove_hooks_if_idle():
    """called when a root model dies: with no live root left (children may outlive their root for a moment, or for good when the
    caller kept one) the map is cleared and the process-wide hooks are removed"""
    if _HOOKS and not any(len(r) for r in list(_TRACKED.values())):
        _TRACKED.clear()
        for h in _HOOKS:
            h.remove()
        del _HOOKS[:]


def _track_tree(root):
    """(re)register every module of `root`'s tree; a submodule that sits in two models (a shared backbone, an init net reachable from
    two PoseNets) is tracked for both"""
    _install_hooks()
    if getattr(root, "_track_fin", None) is None:
        object.__setattr__(root, "_track_fin", weakref.finalize(root, _remove_hooks_if_idle))
    for m in root.modules():
        s_ = _TRACKED.get(m)
        if s_ is None:
            s_ = _TRACKED[m] = weakref.WeakSet()
        s_.add(root)


def _version_sum(tensors):
    try:
        return sum(map(_VERSION_OF, tensors))
    except RuntimeError:        # inference-mode tensors (built / loaded under torch.inference_mode) track no version: they cannot be
        return sum(0 if t.is_inference() else t._version for t in tensors)   # edited in place outside inference mode either
TRAIN_GRAPH = os.environ.get("CHECKERPOSE_AMD_TRAIN_GRAPH", "fwd,bwd").split(",")
# stream priority per graph lane of the eval programs (torch: -1 = high, 0 = default); A/B knob CHECKERPOSE_AMD_LANE_PRIO="-1,0,0"
LANE_PRIO = [int(v) for v in os.environ.get("CHECKERPOSE_AMD_LANE_PRIO", "0").split(",") if v.strip()]   # which halves replay as hipGraphs (A/B + debugging)


def _replay_half(mod, pr, which, lo, hi, device):
    """Run ops [lo, hi) of the training program on the current stream: eagerly the first time (also lets one-time
    kernel attribute setup happen outside a capture), then as one captured hipGraph per half (forward / backward)."""
    prog = pr["prog"]
    cur = torch.cuda.current_stream(device)
    if getattr(prog, "wg_stream", cur.cuda_stream) != cur.cuda_stream:
        raise RuntimeError("this training program was built for stream %#x (its weight-gradient arena is shared with the other "
                           "programs of that stream) and is being replayed on stream %#x: build the model's training programs "
                           "under the stream they run on, or call invalidate()" % (prog.wg_stream, cur.cuda_stream))
    lib = _abi.load()
    g = pr["graphs"].get(which)
    if not mod.use_graph or which[:3] not in TRAIN_GRAPH or not pr["warm"].get(which):
        prog.run_range(cur.cuda_stream, lo, hi)
        pr["warm"][which] = True
        return
    if g is None:
        nl = prog.nlanes if (mod.use_lanes and mod.train_lanes and which == "fwd") else 1
        lanes = [torch.cuda.Stream(device) for _ in range(nl)]
        lanes[0].wait_stream(cur)
        _abi.check(lib.cp_graph_begin_capture(lanes[0].cuda_stream), "graph capture begin")
        try:
            if nl > 1:
                pr["keep_streams"].append(prog.run_lanes_range(lanes, lo, hi))
            elif mod.train_prep_lane:
                lanes.append(torch.cuda.Stream(device))
                pr["keep_streams"].append(prog.run_prep_range(lanes[0], lanes[1], lo, hi))
            else:
                prog.run_range(lanes[0].cuda_stream, lo, hi)
        finally:
            gx = C.c_void_p()
            rc = lib.cp_graph_end_capture(lanes[0].cuda_stream, C.byref(gx))
        _abi.check(rc, "graph capture end")
        pr["graphs"][which] = g = gx
        pr["keep_streams"].append(lanes)
    _abi.check(lib.cp_graph_launch(g, cur.cuda_stream), "graph launch")


def batch_bucket(B):
    """Eval programs (launch list + workspace + hipGraph) are built per batch size; a ragged last batch or a varying number of
    detections would rebuild one per distinct size.  Sizes are therefore rounded up to the next of 1, 2, 3, 4, 6, 8, 12, 16, 24,
    32, 48, 64, ... (2^k and 3*2^(k-1): at most 33 % padding): the B crops run in the first rows of the bucket's program (crops
    are independent -- eval BatchNorm has no cross-sample term, every kernel is per crop), the outputs are sliced back."""
    if B <= 4:
        return B
    k = 1 << (B - 1).bit_length()          # next power of two >= B
    return 3 * k // 4 if 3 * k // 4 >= B else k


class _TrainFn(torch.autograd.Function):
    """autograd hook of the training program (trainer.TrainProgram): forward() replays the forward half of the launch
    list (train-mode BatchNorm, activations saved in the program's workspace), backward() copies the incoming logit /
    seg gradients into the program's fp32 input blocks, replays the backward half and hands autograd the parameter
    gradients (slices of ONE flat fp32 buffer; under torch.distributed that buffer is all-reduced first -- the single
    collective of the data-parallel step)."""

    @staticmethod
    def forward(ctx, mod, pr, img, *params):
        io, prog = pr["io"], pr["prog"]
        io["img"].copy_(img)
        _replay_half(mod, pr, "fwd", 0, prog.n_fwd_ops, img.device)
        if pr["counters"]:
            torch._foreach_add_(pr["counters"], 1)          # BatchNorm2d.num_batches_tracked
        ctx.mod, ctx.pr, ctx.nparams = mod, pr, len(params)
        return io["bits"].clone(), io["seg"].clone() if "seg" in io else io["bits"].new_zeros(1)

    @staticmethod
    def backward(ctx, dbits, dseg):
        pr, mod = ctx.pr, ctx.mod
        io, prog = pr["io"], pr["prog"]
        if pr["busy"] is not ctx:
            raise RuntimeError("checkerpose_amd: backward() of a train-mode forward whose saved activations were overwritten "
                               "by a later forward of the same batch size (one backward per forward)")
        io["dbits"].copy_(dbits)
        nbi = io["dinit"].shape[1]        # 7: rows [roi | x2 x1 x0] + [y2 y1 y0] of the logit block; else an InitNet alone, packed rows
        io["dinit"].copy_(torch.cat([dbits[:, 0:4], dbits[:, 7:10]], dim=1) if nbi == 7 else dbits[:, :nbi])
        if "dseg" in io:
            io["dseg"].copy_(dseg)
        pg = pr["pgrad"]
        # autograd may have stolen last step's views of this buffer as param.grad (zero_grad(set_to_none=False) or gradient
        # accumulation): the backward is about to overwrite it, so such gradients get storage of their own first
        lo_p, hi_p = pg.data_ptr(), pg.data_ptr() + pg.numel() * 4
        for p_ in pr["params"]:
            if p_.grad is not None and lo_p <= p_.grad.data_ptr() < hi_p:
                p_.grad = p_.grad.clone()
        dist_on = (mod.dp_allreduce and torch.distributed.is_available() and torch.distributed.is_initialized()
                   and torch.distributed.get_world_size() > 1)
        if dist_on:       # the ONE exchange of the data-parallel step: bucketed, each bucket overlapped with the rest of the backward
            from ..parallel import backward_with_bucketed_allreduce_
            backward_with_bucketed_allreduce_(pg, pr["segments"],
                                              lambda k, lo, hi: _replay_half(mod, pr, "bwd%d" % k, lo, hi, dbits.device))
        else:
            _replay_half(mod, pr, "bwd", prog.n_fwd_ops, len(prog.calls), dbits.device)
        live = pr["live_params"]      # parameters the (B, stage) program writes; the rest keep grad None (as in the reference)
        grads = tuple(pg[o:o + p.numel()].view_as(p) if (p.requires_grad and lv) else None
                      for (o, p, lv) in zip(pr["offsets"], pr["params"], live))
        pr["busy"] = None
        return (None, None, None) + grads


class HipForwardMixin:
    def _init_runtime(self):
        self._programs = {}
        self._train_programs = {}
        self._stores = {}
        self._idx_dev = None
        self._stale_eval = False          # a train step changed the weights / running stats the eval programs folded
        self.dp_allreduce = True          # all-reduce the flat gradient buffer when torch.distributed is initialised
        self.train_prep_lane = os.environ.get("CHECKERPOSE_AMD_TRAIN_PREP_LANE", "0") == "1"   # weight prep on a side stream
        #                                   under capture: measured -3 % at B=32 (46.5 vs 45.0 ms) -> off
        self.train_lanes = os.environ.get("CHECKERPOSE_AMD_TRAIN_LANES", "0") == "1"   # parallel graph branches in the training
        #                                   forward: measured +-0 % at B=32 (the step is bandwidth-, not latency-bound) -> off
        self.compute_dtype = os.environ.get("CHECKERPOSE_AMD_DTYPE", "fp32")
        self.use_graph = os.environ.get("CHECKERPOSE_AMD_GRAPH", "1") != "0"
        self.use_lanes = os.environ.get("CHECKERPOSE_AMD_LANES", "1") != "0"   # parallel graph branches
        self.max_lanes = int(os.environ.get("CHECKERPOSE_AMD_MAX_LANES", "0"))    # 0 = as many streams as the program has lanes (A/B knob)
        self.use_dag = os.environ.get("CHECKERPOSE_AMD_DAG", "0") == "1"       # dataflow capture (Program.run_dag) instead of fork/join lanes
        self.batch_splits = int(os.environ.get("CHECKERPOSE_AMD_SPLITS", "1"))   # concurrent batch slices per forward (measured: 1 is fastest; 2 and 4 lose 8 % / 30 % at B=128)
        self.clone_outputs = True
        self.kernel_selection = os.environ.get("CHECKERPOSE_AMD_SELECTION", "auto")
        self.batch_buckets = os.environ.get("CHECKERPOSE_AMD_BUCKETS", "1") != "0"   # eval: pad ragged batches to a cached size
        self.check_weight_versions = os.environ.get("CHECKERPOSE_AMD_CHECK_VERSIONS", "1") != "0"   # eval: detect in-place weight edits
        self._sig_tensors, self._sig_epoch, self._struct_epoch = None, -1, 0
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate())

    # ---- cache control
    def check_device_status(self):
        """Raise if a kernel reported a failure of its own (include/checkerpose_hip.h: cp_device_status) since the last look.
        Synchronises the device.  Called by the model wherever it synchronises anyway (program build, invalidate()); call it
        yourself behind a batch of forwards whose results matter."""
        with torch.cuda.device(next(self.parameters()).device):
            _abi.raise_on_device_status(type(self).__name__)

    def invalidate(self):
        """Drop packed weights / programs (called on load_state_dict, .to()/.cuda(), dtype change)."""
        lib = _abi._lib
        had_programs = bool(getattr(self, "_programs", None) or getattr(self, "_train_programs", None))
        for pr in getattr(self, "_programs", {}).values():
            if pr.get("graph") and lib is not None:
                for g in pr["graph"]:
                    lib.cp_graph_destroy(g)
        for pr in getattr(self, "_train_programs", {}).values():
            for g in pr["graphs"].values():
                if lib is not None:
                    lib.cp_graph_destroy(g)
        self._programs, self._stores, self._idx_dev = {}, {}, None
        self._train_programs = {}
        self._sig_tensors = None
        self._tiled = None
        p0 = next(self.parameters(), None)
        if had_programs and lib is not None and p0 is not None and p0.is_cuda:     # the programs that ran: did one of their kernels report a failure?
            with torch.cuda.device(p0.device):
                _abi.raise_on_device_status("%s.invalidate()" % type(self).__name__)
        # the owning PoseNet folded this init net's weights into ITS programs, and vice versa: drop those too
        for other in (getattr(self, "_owner", None), getattr(self, "init_net", None)):
            other = other() if callable(other) and not isinstance(other, torch.nn.Module) else other
            if other is not None and hasattr(other, "_programs") and (other._programs or other._train_programs or other._stores):
                other.invalidate()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if hasattr(self, "_programs"):
            self.invalidate()
        return r

    def set_kernel_selection(self, mode):
        """Which launches a bf16 program is made of depends on the batch size ("auto": per-conv / split-K launches for small
        batches, per-crop LDS-resident launches from 40-96 crops: the measured crossovers), and the two selections accumulate in
        a different order -- the same crop gives different bf16 bits at batch 32 and batch 64.  "per_crop" / "tiled" pin ONE
        selection for every batch size (bit-reproducible across batch sizes, at the price of the other regime's speed)."""
        if mode not in ("auto", "per_crop", "tiled"):
            raise ValueError("kernel selection must be 'auto', 'per_crop' or 'tiled'")
        self.kernel_selection = mode
        child = getattr(self, "init_net", None)
        if child is not None and hasattr(child, "kernel_selection"):
            child.kernel_selection = mode
            child.invalidate()
        self.invalidate()
        return self

    @property
    def deterministic(self):
        """deterministic training mode (checkerpose_amd.set_deterministic): process-wide, readable / settable on any drop-in model"""
        return bool(_abi.load().cp_get_deterministic())

    @deterministic.setter
    def deterministic(self, on):
        lib = _abi.load()
        if bool(lib.cp_get_deterministic()) != bool(on):
            lib.cp_set_deterministic(1 if on else 0)
            self.invalidate()              # launch plans (block counts, accumulator sets, reduction paths) depend on the mode

    def set_compute_dtype(self, name):
        if name not in DTYPES:
            raise ValueError("compute dtype must be one of %s" % sorted(DTYPES))
        self.compute_dtype = name
        self.invalidate()
        return self

    # ---- program construction
    def _build(self, lib, B, size, stage, want_feats, want_graph, device, teacher=False, u8=False, inject=False, inject_dec=False,
               want_dec=False):
        dtype = DTYPES[self.compute_dtype]
        cfg = self._net_cfg()
        cfg["img_size"] = size
        cfg["stage"] = stage
        cfg["uint8_input"] = u8
        cfg["inject_feats"] = inject
        cfg["inject_dec"], cfg["export_dec"] = inject_dec, want_dec
        N = cfg["npoint"]
        sd = self.state_dict()
        if dtype not in self._stores:
            self._stores[dtype] = WeightStore(lib, sd, dtype, device)
        ws = self._stores[dtype]
        ws.sd = sd
        if self._idx_dev is None:
            self._idx_dev = self._knn_table().to(torch.int32).contiguous().to(device)
        idx = self._idx_dev
        G = idx.shape[0]
        tiled = self._tile_tables(lib, B, N, idx, device) if (dtype == DTYPES["bf16"] and not want_graph) else None
        io = dict(
            img=(torch.empty(B, size, size, 3, dtype=torch.uint8, device=device) if u8
                 else torch.empty(B, 3, size, size, dtype=torch.float32, device=device)),
            bits=torch.zeros(B, 13, N, dtype=torch.float32, device=device),
            mask=torch.zeros(B, N, dtype=torch.float32, device=device),
            xid=torch.zeros(B, N, dtype=torch.int32, device=device),
            yid=torch.zeros(B, N, dtype=torch.int32, device=device),
            x64=torch.zeros(B, N, dtype=torch.int64, device=device),
            y64=torch.zeros(B, N, dtype=torch.int64, device=device),
            gids=torch.zeros(B, dtype=torch.int32, device=device) if self.LM else None,
        )
        if teacher:   # teacher forcing (tests): the discrete feedback is decoded from THESE logits, not the computed ones
            io["decode_bits"] = torch.zeros(B, 13, N, dtype=torch.float32, device=device)
        if inject:    # injected backbone features (tests): NCHW fp32, the layout the reference's backbone returns them in
            from ..netbuilder import IMG_FEATS_DIMS as _FD
            io["inject"] = [torch.zeros(B, c, size >> (2 + i), size >> (2 + i), dtype=torch.float32, device=device)
                            for i, c in enumerate(_FD[cfg["backbone"]])]
        if inject_dec or want_dec:      # attribution hook (agreement.attribute_groups): the decoder stages' output maps, NCHW fp32
            nst = stage if stage is not None else cfg["res_log2"] - 3
            mk = lambda: [torch.zeros(B, cfg["num_filters"], size >> (4 - i), size >> (4 - i), dtype=torch.float32, device=device)   # noqa: E731
                          for i in range(nst)]
            if inject_dec:
                io["inject_dec"] = mk()
            if want_dec:
                io["dec_feats"] = mk()
        # batch slices: independent sub-programs that the captured graph runs concurrently (see ProgramGroup)
        nsplit = self.batch_splits if (B >= 8 * self.batch_splits and B % self.batch_splits == 0) else 1
        # the kernels address every tensor through 32-bit buffer descriptors (< 2 GiB): the widest one is the last decoder
        # stage's input concat (size/4)^2 x (num_filters + skip channels) -> fp32 batches beyond 255 crops run as slices
        from ..netbuilder import IMG_FEATS_DIMS
        widest = (size // 4) ** 2 * (cfg.get("num_filters", 0) + IMG_FEATS_DIMS[cfg["backbone"]][1]) * (4 if self.compute_dtype in ("fp32", "f32", "float32") else 2)
        need = -(-B * widest // ((1 << 31) - 1))
        if need > nsplit:
            nsplit = need
            while B % nsplit:
                nsplit += 1
        Bs = B // nsplit
        fs = None
        if cfg["kind"] != "init":
            nref = cfg["res_log2"] - 3
            fs = (size // 32) << (stage if stage is not None else nref)
            io["seg"] = torch.zeros(B, cfg["seg_output_dim"], fs, fs, dtype=torch.float32, device=device)
        if want_feats:
            io["img_feats"] = None
        progs = []
        for si in range(nsplit):
            sl = slice(si * Bs, (si + 1) * Bs)
            sio = {k: (v[sl] if torch.is_tensor(v) else ([t[sl] for t in v] if isinstance(v, list) else v)) for k, v in io.items()}
            prog = Program(lib, ws, dtype, Bs, device)
            if self.kernel_selection == "per_crop":
                prog.chain_min = prog.stem_min = prog.edge_min = prog.mlp_min_rows = prog.fuse_out_min = 1
                prog.splitk = False
            elif self.kernel_selection == "tiled":
                prog.chain_min = prog.stem_min = prog.edge_min = prog.mlp_min_rows = prog.fuse_out_min = 1 << 30
                prog.splitk = False
            sio["graph"] = dict(idx=idx, gids=sio["gids"], K=idx.shape[2], G=G)
            # the per-keypoint block group in IEEE half when each of its ops takes the fused per-crop kernel at this batch (engine.USE_GNN_F16)
            prog.gnn_half = (not want_graph) and prog.wants_gnn_half(N, idx.shape[2], tiled["HPAD"] if tiled is not None else None)
            ext = None
            if tiled is not None and prog.can_tile_edgeconv(N, idx.shape[2], 64, 64, tiled["HPAD"]):
                # large graphs: the program works in the patch-ordered INTERNAL keypoint numbering (graph_sched.tile_schedule): its
                # logit block / ids are internal tensors, un-permuted into the caller-visible ones behind the last launch
                sio["graph"].update(tiled=tiled, idx=tiled["idx_internal"])
                ext = {k: sio[k] for k in ("bits", "x64", "y64")}
                for k in ext:
                    sio[k] = torch.zeros_like(ext[k])
                if teacher:
                    tb = sio["decode_bits"]
                    sio["decode_bits"] = torch.zeros_like(tb)
                    prog.permute_cols(tb, sio["decode_bits"], tiled["perm"], sio["gids"], 13, N, scatter=False)
            sio["bits_tb"] = prog.fixed(sio["bits"])
            em = NetEmitter(prog, sd)
            if cfg["kind"] == "init":
                feats, g = emit_init_net(em, cfg, sio, "")
            else:
                sio["seg_tb"] = prog.fixed(sio["seg"])
                feats, _ = emit_posenet(em, cfg, sio)
                g = None
            if ext is not None:
                prog.permute_cols(sio["bits"], ext["bits"], tiled["perm"], sio["gids"], 13, N, scatter=True)
                prog.permute_cols(sio["x64"], ext["x64"], tiled["perm"], sio["gids"], 1, N, scatter=True)
                prog.permute_cols(sio["y64"], ext["y64"], tiled["perm"], sio["gids"], 1, N, scatter=True)
            if want_feats:
                if io["img_feats"] is None:
                    io["img_feats"] = [torch.empty(B, f.C, f.H, f.W, dtype=torch.float32, device=device) for f in feats]
                for f, t in zip(feats, io["img_feats"]):
                    prog.to_nchw_f32(f, t[sl])
            if want_graph and g is not None:
                if "graph_feats" not in io:
                    io["graph_feats"] = torch.empty(B, g.C, N, dtype=torch.float32, device=device)
                prog.to_nchw_f32(g, io["graph_feats"][sl])
            prog.finalize(dce=os.environ.get("CHECKERPOSE_AMD_DCE", "1") != "0")       # e.g. timm's highest-resolution incre feature: computed by the reference, read by nobody
            progs.append(prog)
        prog = ProgramGroup(progs)
        torch.cuda.current_stream(device).synchronize()      # weight packing done before temporaries die
        ws.keep.clear()
        _abi.raise_on_device_status("building a program")   # also creates the device's status word BEFORE any launch / capture
        return dict(prog=prog, io=io, graph=None, warm=False, side=None)

    def _tile_tables(self, lib, B, N, idx_dev, device):
        """patch schedule of a large graph (N > 512) as device tensors, built once per module (init-time, host: ~1 s per graph)"""
        from .. import engine
        if not engine.USE_EDGE_TILED or N <= 512 or N % 512:
            return None
        if getattr(self, "_tiled", None) is None:
            if getattr(self, "_tile_host", None) is None:        # the graph never changes: the host schedule survives invalidate()
                from ..graph_sched import tile_schedule
                pts = self._keypoints() if hasattr(self, "_keypoints") else None
                self._tile_host = (tile_schedule(self._knn_table().cpu().numpy(), pts.numpy()) if pts is not None else None) or False
            sc = self._tile_host or None
            if sc is None:
                self._tiled = False
            else:
                self._tiled = dict(HPAD=sc["HPAD"], NB=sc["NB"], halo_rows=sc["halo_rows"],
                                   **{k: torch.from_numpy(sc[k]).contiguous().to(device) for k in ("perm", "halo", "nbr", "idx_internal")})
        return self._tiled or None

    # ---- training program (forward in train mode + backward), see ../trainer.py
    def _build_train(self, lib, B, size, stage, device, u8=False):
        from ..trainer import TrainProgram, TrainWeightStore
        from ..train_ops import reverse_graph
        dtype = DTYPES[self.compute_dtype]
        cfg = self._net_cfg()
        cfg["img_size"], cfg["stage"], cfg["uint8_input"] = size, stage, u8
        N = cfg["npoint"]
        sd = self.state_dict()
        params, offsets, slots, off = [], [], {}, 0
        for name, p in self.named_parameters():
            if p.dtype != torch.float32 or p.device != device:
                raise RuntimeError("parameter %s must be an fp32 tensor on %s" % (name, device))
            params.append(p)
            offsets.append(off)
            slots[name] = off
            off += (p.numel() + 3) // 4 * 4
        pgrad = torch.zeros(off, dtype=torch.float32, device=device)
        ws = TrainWeightStore(lib, sd, dtype, device)
        if self._idx_dev is None:
            self._idx_dev = self._knn_table().to(torch.int32).contiguous().to(device)
        idx = self._idx_dev
        rev_ptr, rev_edge = reverse_graph(idx)
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)   # noqa: E731
        io = dict(img=(z(B, size, size, 3, dt=torch.uint8) if u8 else z(B, 3, size, size)), bits=z(B, 13, N), mask=z(B, N), xid=z(B, N, dt=torch.int32), yid=z(B, N, dt=torch.int32),
                  x64=z(B, N, dt=torch.int64), y64=z(B, N, dt=torch.int64), gids=z(B, dt=torch.int32) if self.LM else None,
                  dbits=z(B, 13, N), dinit=z(B, 1 + 2 * cfg.get("init_res_log2", 3), N))
        if cfg["kind"] != "init":
            nref = cfg["res_log2"] - 3
            fs = (size // 32) << (stage if stage is not None else nref)
            io["seg"] = z(B, cfg["seg_output_dim"], fs, fs)
            io["dseg"] = z(B, cfg["seg_output_dim"], fs, fs)
        prog = TrainProgram(lib, ws, dtype, B, device, pgrad, slots)
        io["graph"] = dict(idx=idx, gids=io["gids"], K=idx.shape[2], G=idx.shape[0], rev_ptr=rev_ptr, rev_edge=rev_edge)
        io["bits_tb"] = prog.fixed(io["bits"])
        em = NetEmitter(prog, sd)
        if cfg["kind"] == "init":
            emit_init_net(em, cfg, io, "")
        else:
            io["seg_tb"] = prog.fixed(io["seg"])
            emit_posenet(em, cfg, io)
        prog.mark_forward_end()
        nb = pgrad.numel() * 4
        prog._add(lib.cp_memset_zero, lambda P: (pgrad.data_ptr(), nb), "pgrad_zero", [], [])
        prog.zero_grad_arena()
        prog.unwind()
        prog.finalize()
        torch.cuda.current_stream(device).synchronize()
        names = [n for n, _ in self.named_parameters()]
        return dict(prog=prog, io=io, pgrad=pgrad, params=params, offsets=offsets, counters=list(em.bn_counters), busy=None,
                    graphs={}, warm={}, keep_streams=[], ptrs=self._storage_signature(),
                    segments=prog.gradient_buckets(int(os.environ.get("CHECKERPOSE_AMD_GRAD_BUCKETS", "4"))),
                    live_params=[n in prog.pslot_done for n in names])

    def _storage_signature(self):
        """the launch program holds raw pointers into the parameter / buffer storage (live weights): if anything re-assigned a
        tensor (`p.data = ...`, a non-in-place optimizer) the program must be rebuilt"""
        return tuple(t.data_ptr() for t in self.parameters()) + tuple(t.data_ptr() for t in self.buffers())

    def _run_train(self, img, obj_ids, stage=None):
        if not (torch.is_tensor(img) and img.is_cuda):
            raise RuntimeError("checkerpose_amd: input must be a CUDA/HIP tensor on an MI355X; there is no CPU fallback.")
        u8 = img.dtype == torch.uint8       # raw (B,256,256,3) uint8 crops, normalised on the device (row N3), as in eval mode
        if u8:
            if img.dim() != 4 or img.shape[3] != 3 or img.shape[1] != img.shape[2] or img.shape[1] != 256:
                raise ValueError("expected uint8 img of shape (B, 256, 256, 3), got %s" % (tuple(img.shape),))
        elif img.dim() != 4 or img.shape[1] != 3 or img.shape[2] != img.shape[3] or img.shape[2] != 256 or img.dtype != torch.float32:
            raise ValueError("expected fp32 img of shape (B, 3, 256, 256), got %s" % (tuple(img.shape),))
        lib = _abi.load()
        device = img.device
        B, size = img.shape[0], img.shape[2]
        key = (B, size, stage, self.compute_dtype, u8, bool(lib.cp_get_deterministic()))
        pr = self._train_programs.get(key)
        if pr is not None and pr["ptrs"] != self._storage_signature():
            for g in pr["graphs"].values():
                lib.cp_graph_destroy(g)
            pr = None
        if pr is None:
            dist = torch.distributed
            if (self.dp_allreduce and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                    and not getattr(self, "_dp_synced", False)):
                # data-parallel replicas must start from identical parameters AND BatchNorm buffers: rank 0's are broadcast
                # once (replicas that were not seeded identically would otherwise diverge silently)
                for t in list(self.parameters()) + list(self.buffers()):
                    dist.broadcast(t.data, 0)
                self._dp_synced = True
            with torch.cuda.device(device):
                pr = self._build_train(lib, B, size, stage, device, u8)
            self._train_programs[key] = pr
        io = pr["io"]
        self._stale_eval = True                 # ... of this module, of the init net inside it and of the PoseNet around it
        for other in (getattr(self, "_owner", None), getattr(self, "init_net", None)):
            other = other() if callable(other) and not isinstance(other, torch.nn.Module) else other
            if other is not None and hasattr(other, "_stale_eval"):
                other._stale_eval = True
        with torch.cuda.device(device):
            if self.LM:
                if obj_ids is None:
                    raise ValueError("obj_ids is required for the LM networks")
                io["gids"].copy_((obj_ids.to(device) - 1).to(torch.int32))
            if torch.is_grad_enabled():
                bits, seg = _TrainFn.apply(self, pr, img, *pr["params"])
                pr["busy"] = bits.grad_fn
            else:                                     # train-mode statistics without autograd (e.g. under no_grad)
                io["img"].copy_(img)
                _replay_half(self, pr, "fwd", 0, pr["prog"].n_fwd_ops, device)
                if pr["counters"]:
                    torch._foreach_add_(pr["counters"], 1)
                bits, seg = io["bits"].clone(), io["seg"].clone() if "seg" in io else None
        out = dict(bits=bits, x64=io["x64"].clone(), y64=io["y64"].clone())
        if "seg" in io:
            out["seg"] = seg
        return out

    def train(self, mode=True):
        r = super().train(mode)
        if not mode and getattr(self, "_stale_eval", False):     # eval programs fold BatchNorm / pack weights at build time
            self._drop_eval_programs()
        return r

    def _drop_eval_programs(self):
        """the eval programs (folded BatchNorm, packed weights, graphs) are stale; the training programs read live weights"""
        lib = _abi._lib
        for pr in self._programs.values():
            if pr.get("graph") and lib is not None:
                for g in pr["graph"]:
                    lib.cp_graph_destroy(g)
        self._programs, self._stores = {}, {}
        self._stale_eval = False

    # ---- one forward
    def _run(self, img, obj_ids, stage=None, want_feats=False, want_graph=False, teacher_bits=None, inject_feats=None,
             inject_dec=None, want_dec=False):
        if self.training:
            if want_feats or want_graph or teacher_bits is not None or inject_feats is not None or inject_dec is not None or want_dec:
                raise RuntimeError("checkerpose_amd: return_img_feats / return_graph_feats / teacher forcing are eval-mode only")
            return self._run_train(img, obj_ids, stage=stage)
        if not (torch.is_tensor(img) and img.is_cuda):
            raise RuntimeError("checkerpose_amd: input must be a CUDA/HIP tensor on an MI355X; there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only).")
        u8 = img.dtype == torch.uint8       # extension: raw (B,256,256,3) uint8 crops, normalised on the device
        if u8:
            if img.dim() != 4 or img.shape[3] != 3 or img.shape[1] != img.shape[2] or img.shape[1] != 256:
                raise ValueError("expected uint8 img of shape (B, 256, 256, 3), got %s" % (tuple(img.shape),))
        elif img.dim() != 4 or img.shape[1] != 3 or img.shape[2] != img.shape[3] or img.shape[2] != 256:
            raise ValueError("expected img of shape (B, 3, 256, 256), got %s" % (tuple(img.shape),))
        lib = _abi.load()
        device = img.device
        p0 = next(self.parameters())
        if p0.device != device:
            raise RuntimeError("module parameters are on %s but the input is on %s" % (p0.device, device))
        Bu, size = img.shape[0], img.shape[2]                 # the caller's batch; B: the (possibly larger) cached program size
        B = batch_bucket(Bu) if self.batch_buckets else Bu
        if self.check_weight_versions:
            # eval programs fold BatchNorm and pack weights at build time: an in-place edit of any parameter / buffer since then
            # (optimizer step, EMA `copy_` / `mul_` under no_grad, a child's load_state_dict) bumps its version counter -> rebuild.
            # (Edits through `p.data` do not move the counter -- torch gives `.data` a counter of its own: call invalidate().)
            if self._sig_tensors is None or self._sig_epoch != self._struct_epoch:
                _track_tree(self)
                ts = list(self.parameters()) + list(self.buffers())
                if self._sig_tensors is not None and (len(ts) != len(self._sig_tensors) or
                                                      any(a is not b for a, b in zip(ts, self._sig_tensors))):
                    self._eval_sig = None                    # other tensor objects than the programs folded: rebuild
                self._sig_tensors, self._sig_epoch = ts, self._struct_epoch
            sig = _version_sum(self._sig_tensors) + len(self._sig_tensors)      # ~2 000 tensors: the B = 1 forward is host-bound
            if self._programs and sig != getattr(self, "_eval_sig", None):
                self._drop_eval_programs()
            self._eval_sig = sig
        key = (B, size, stage, self.compute_dtype, want_feats, want_graph, teacher_bits is not None, u8, inject_feats is not None,
               inject_dec is not None, want_dec)
        pr = self._programs.get(key)
        if pr is None:
            if self.check_weight_versions:
                _track_tree(self)               # (also covers a model that was deep-copied together with its epoch)
            with torch.cuda.device(device):
                pr = self._build(lib, B, size, stage, want_feats, want_graph, device, teacher_bits is not None, u8,
                                 inject_feats is not None, inject_dec is not None, want_dec)
            self._programs[key] = pr
        io, prog = pr["io"], pr["prog"]
        with torch.cuda.device(device):
            if img.data_ptr() != io["img"].data_ptr():        # zero-copy when the caller filled input_buffer(B)
                io["img"][:Bu].copy_(img)                     # boundary: stage the caller's NCHW fp32 batch
            if teacher_bits is not None:
                io["decode_bits"][:Bu].copy_(teacher_bits)
            if inject_feats is not None:
                for dst, src in zip(io["inject"], inject_feats):
                    dst[:Bu].copy_(src)
            if inject_dec is not None:
                for dst, src in zip(io["inject_dec"], inject_dec):
                    dst[:Bu].copy_(src)
            if self.LM:
                if obj_ids is None:
                    raise ValueError("obj_ids is required for the LM networks")
                io["gids"][:Bu].copy_((obj_ids.to(device) - 1).to(torch.int32))   # obj_ids start from 1 (init_lm.py:65)
            cur = torch.cuda.current_stream(device)
            if self.use_graph and pr["warm"]:
                if pr["graph"] is None:
                    # one hipGraph per batch slice (each with its own parallel lanes); the slices' graphs are then
                    # launched concurrently on separate streams, forked from / joined to the caller's stream
                    graphs, keep = [], []
                    for sub in prog.progs:
                        nl_ = sub.nlanes if self.use_lanes and not self.use_dag else 1
                        # lane priorities (they carry into the captured graph): lane 0 = the 64x64 HRNet branch, whose launch is the
                        # longest of a module AND has the longest dependent tail (its 2nd / 3rd stride-2 fuse convs); see LANE_PRIO
                        lanes = [torch.cuda.Stream(device, priority=(LANE_PRIO[k] if k < len(LANE_PRIO) else 0)) for k in range(nl_)]
                        if self.max_lanes and len(lanes) > self.max_lanes:      # lane k -> stream k mod max_lanes (fewer queues in the graph)
                            lanes = [lanes[k % self.max_lanes] for k in range(len(lanes))]
                        lanes[0].wait_stream(cur)
                        _abi.check(lib.cp_graph_begin_capture(lanes[0].cuda_stream), "graph capture begin")
                        try:
                            if self.use_dag:
                                sub.run_dag(lanes[0].cuda_stream)
                            elif len(lanes) > 1:
                                keep.append(sub.run_lanes(lanes))
                            else:
                                sub.run(lanes[0].cuda_stream)
                        finally:
                            gx = C.c_void_p()
                            rc = lib.cp_graph_end_capture(lanes[0].cuda_stream, C.byref(gx))
                        _abi.check(rc, "graph capture end")
                        graphs.append(gx)
                        keep.append(lanes)
                    pr["graph"] = graphs
                    pr["side"] = dict(keep=keep, streams=[torch.cuda.Stream(device) for _ in graphs[1:]],
                                      fork=torch.cuda.Event(), joins=[torch.cuda.Event() for _ in graphs[1:]])
                graphs, sd_ = pr["graph"], pr["side"]
                if len(graphs) > 1:
                    sd_["fork"].record(cur)
                    for g, st_, ej in zip(graphs[1:], sd_["streams"], sd_["joins"]):
                        st_.wait_event(sd_["fork"])
                        _abi.check(lib.cp_graph_launch(g, st_.cuda_stream), "graph launch")
                        ej.record(st_)
                _abi.check(lib.cp_graph_launch(graphs[0], cur.cuda_stream), "graph launch")
                for ej in (sd_["joins"] if len(graphs) > 1 else []):
                    cur.wait_event(ej)
            else:
                prog.run(cur.cuda_stream)
                pr["warm"] = True
        out = {k: io[k] for k in ("bits", "seg", "x64", "y64", "img_feats", "graph_feats", "dec_feats") if k in io}
        if Bu != B:
            out = {k: ([t[:Bu] for t in v] if isinstance(v, list) else v[:Bu]) for k, v in out.items()}
        if self.clone_outputs:
            out = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}
        return out

    def input_buffer(self, B, stage=None):
        """The persistent (B,3,256,256) fp32 device tensor the launch program reads.  A producer (data loader, bench)
        that writes crops into it and passes it to forward() skips the staging copy.  Needs one prior forward at B."""
        Bp = batch_bucket(B) if self.batch_buckets else B
        for k, pr in self._programs.items():
            if k[0] == Bp and k[2] == stage and k[3] == self.compute_dtype and not k[7]:
                return pr["io"]["img"][:B]
        raise RuntimeError("no program for batch %d yet: run one forward first" % B)

    def program_for(self, B, stage=None):
        """Introspection for bench / tests: the cached Program of batch size B (after at least one forward)."""
        Bp = batch_bucket(B) if self.batch_buckets else B
        for k, pr in self._programs.items():
            if k[0] == Bp and k[2] == stage and k[3] == self.compute_dtype:
                return pr["prog"]
        return None
