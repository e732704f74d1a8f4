"""Runtime glue shared by the drop-in modules: builds / caches the launch Program for a batch size, owns the
persistent I/O tensors, captures the hipGraph and replays it.  Fails loudly -- no CPU or PyTorch fallback.
"""
import ctypes as C
import os

import torch

from .. import _abi
from ..engine import DTYPES, Program, ProgramGroup, WeightStore
from ..netbuilder import NetEmitter, emit_init_net, emit_posenet


class HipForwardMixin:
    def _init_runtime(self):
        self._programs = {}
        self._stores = {}
        self._idx_dev = None
        self.compute_dtype = os.environ.get("CHECKERPOSE_AMD_DTYPE", "fp32")
        self.use_graph = os.environ.get("CHECKERPOSE_AMD_GRAPH", "1") != "0"
        self.use_lanes = os.environ.get("CHECKERPOSE_AMD_LANES", "1") != "0"   # parallel graph branches
        self.batch_splits = int(os.environ.get("CHECKERPOSE_AMD_SPLITS", "1"))   # concurrent batch slices per forward (measured: 1 is fastest; 2 and 4 lose 8 % / 30 % at B=128)
        self.clone_outputs = True
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate())

    # ---- cache control
    def invalidate(self):
        """Drop packed weights / programs (called on load_state_dict, .to()/.cuda(), dtype change)."""
        lib = _abi._lib
        for pr in getattr(self, "_programs", {}).values():
            if pr.get("graph") and lib is not None:
                for g in pr["graph"]:
                    lib.cp_graph_destroy(g)
        self._programs, self._stores, self._idx_dev = {}, {}, None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if hasattr(self, "_programs"):
            self.invalidate()
        return r

    def set_compute_dtype(self, name):
        if name not in DTYPES:
            raise ValueError("compute dtype must be one of %s" % sorted(DTYPES))
        self.compute_dtype = name
        self.invalidate()
        return self

    # ---- program construction
    def _build(self, lib, B, size, stage, want_feats, want_graph, device, teacher=False, u8=False):
        dtype = DTYPES[self.compute_dtype]
        cfg = self._net_cfg()
        cfg["img_size"] = size
        cfg["stage"] = stage
        cfg["uint8_input"] = u8
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
        # batch slices: independent sub-programs that the captured graph runs concurrently (see ProgramGroup)
        nsplit = self.batch_splits if (B >= 8 * self.batch_splits and B % self.batch_splits == 0) else 1
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
            sio = {k: (v[sl] if torch.is_tensor(v) else v) for k, v in io.items()}
            prog = Program(lib, ws, dtype, Bs, device)
            sio["graph"] = dict(idx=idx, gids=sio["gids"], K=idx.shape[2], G=G)
            sio["bits_tb"] = prog.fixed(sio["bits"])
            em = NetEmitter(prog, sd)
            if cfg["kind"] == "init":
                feats, g = emit_init_net(em, cfg, sio, "")
            else:
                sio["seg_tb"] = prog.fixed(sio["seg"])
                feats, _ = emit_posenet(em, cfg, sio)
                g = None
            if want_feats:
                if io["img_feats"] is None:
                    io["img_feats"] = [torch.empty(B, f.C, f.H, f.W, dtype=torch.float32, device=device) for f in feats]
                for f, t in zip(feats, io["img_feats"]):
                    prog.to_nchw_f32(f, t[sl])
            if want_graph and g is not None:
                if "graph_feats" not in io:
                    io["graph_feats"] = torch.empty(B, g.C, N, dtype=torch.float32, device=device)
                prog.to_nchw_f32(g, io["graph_feats"][sl])
            prog.finalize()
            progs.append(prog)
        prog = ProgramGroup(progs)
        torch.cuda.current_stream(device).synchronize()      # weight packing done before temporaries die
        ws.keep.clear()
        return dict(prog=prog, io=io, graph=None, warm=False, side=None)

    # ---- one forward
    def _run(self, img, obj_ids, stage=None, want_feats=False, want_graph=False, teacher_bits=None):
        if self.training:
            raise RuntimeError("checkerpose_amd: train-mode forward (batch-statistics BatchNorm + autograd) is not "
                               "implemented; call .eval().  There is no PyTorch fallback.")
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
        B, size = img.shape[0], img.shape[2]
        key = (B, size, stage, self.compute_dtype, want_feats, want_graph, teacher_bits is not None, u8)
        pr = self._programs.get(key)
        if pr is None:
            with torch.cuda.device(device):
                pr = self._build(lib, B, size, stage, want_feats, want_graph, device, teacher_bits is not None, u8)
            self._programs[key] = pr
        io, prog = pr["io"], pr["prog"]
        with torch.cuda.device(device):
            if img.data_ptr() != io["img"].data_ptr():        # zero-copy when the caller filled input_buffer(B)
                io["img"].copy_(img)                          # boundary: stage the caller's NCHW fp32 batch
            if teacher_bits is not None:
                io["decode_bits"].copy_(teacher_bits)
            if self.LM:
                if obj_ids is None:
                    raise ValueError("obj_ids is required for the LM networks")
                io["gids"].copy_((obj_ids.to(device) - 1).to(torch.int32))   # obj_ids start from 1 (init_lm.py:65)
            cur = torch.cuda.current_stream(device)
            if self.use_graph and pr["warm"]:
                if pr["graph"] is None:
                    # one hipGraph per batch slice (each with its own parallel lanes); the slices' graphs are then
                    # launched concurrently on separate streams, forked from / joined to the caller's stream
                    graphs, keep = [], []
                    for sub in prog.progs:
                        lanes = [torch.cuda.Stream(device) for _ in range(sub.nlanes if self.use_lanes else 1)]
                        lanes[0].wait_stream(cur)
                        _abi.check(lib.cp_graph_begin_capture(lanes[0].cuda_stream), "graph capture begin")
                        try:
                            if len(lanes) > 1:
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
        out = {k: io[k] for k in ("bits", "seg", "x64", "y64", "img_feats", "graph_feats") if k in io}
        if self.clone_outputs:
            out = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}
        return out

    def input_buffer(self, B, stage=None):
        """The persistent (B,3,256,256) fp32 device tensor the launch program reads.  A producer (data loader, bench)
        that writes crops into it and passes it to forward() skips the staging copy.  Needs one prior forward at B."""
        for k, pr in self._programs.items():
            if k[0] == B and k[2] == stage and k[3] == self.compute_dtype:
                return pr["io"]["img"]
        raise RuntimeError("no program for batch %d yet: run one forward first" % B)

    def program_for(self, B, stage=None):
        """Introspection for bench / tests: the cached Program of batch size B (after at least one forward)."""
        for k, pr in self._programs.items():
            if k[0] == B and k[2] == stage and k[3] == self.compute_dtype:
                return pr["prog"]
        return None
