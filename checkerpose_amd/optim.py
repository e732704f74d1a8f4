"""Optimizers of the training loop as ONE HIP launch per step (`csrc/optim.hip`).

The reference builds `optim.Adam(net.parameters(), lr=learning_rate)` or `optim.SGD(net.parameters(), lr=..., momentum=0.9)`
(`train.py:244-246`, `train_lm.py:218-220`, `pretrain*.py`) and calls `optimizer.zero_grad()` / `optimizer.step()` once per batch
(`train.py:303,320`).  `Adam` / `SGD` here take the same constructor arguments and are `torch.optim.Optimizer` subclasses
(param groups, `zero_grad`, `state_dict` / `load_state_dict` with torch's state keys `step`, `exp_avg`, `exp_avg_sq` /
`momentum_buffer`), but a step is a single `cp_adam_multi` / `cp_sgd_multi` launch over a device table of all parameter tensors
instead of torch's ~30 multi-tensor launches and ~10 ms of host work per step for the ~1 000 tensors of PoseNet_GNNskip -- the
training step replays its forward / backward as hipGraphs and had become bound by how fast the host enqueues it
(`tools/train_cpu_timeline.py`).  fp32 parameters on one GPU; anything else raises (no silent fallback)."""
import torch

from . import _abi
from ._abi import CpOptItem


class _MultiTensorOptimizer(torch.optim.Optimizer):
    _state_keys = ()

    def _table(self, group, params):
        """device table of (param, grad, state...) of the parameters that have a gradient; cached while no pointer moves"""
        sig = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
        cache = group.setdefault("_cp_table", {})
        if cache.get("sig") == sig:
            return cache
        dev = params[0].device
        items, nbs = [], []
        lib = _abi.load()
        for p in params:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() \
                    or p.device != dev or dev.type != "cuda" or p.grad.is_sparse:
                raise RuntimeError("checkerpose_amd.optim: contiguous fp32 parameters and gradients on one GPU only")
            st = self.state[p]
            if not st:
                self._init_state(p, st)
                self._step0[p] = group.get("_cp_step", 0)       # torch counts steps per parameter: this one starts now
            it = CpOptItem()
            it.p, it.g, it.n, it.step0 = p.data_ptr(), p.grad.data_ptr(), p.numel(), self._step0.get(p, 0)
            self._fill_state_ptrs(it, st)
            items.append(it)
            nbs.append(lib.cp_opt_item_blocks(p.numel()))
        raw, prefix, total = _abi.device_table(items, nbs, dev)
        cache.update(sig=sig, n=len(items), blocks=total, raw=raw, prefix=prefix)
        return cache

    def state_dict(self):
        for g in self.param_groups:              # torch's per-parameter step counters, refreshed only when somebody looks
            for p in g["params"]:
                if p in self.state and "step" in self.state[p]:
                    self.state[p]["step"].fill_(g.get("_cp_step", 0) - self._step0.get(p, 0))
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_cp_table", None)
            g.pop("_cp_step", None)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g.pop("_cp_table", None)              # the loaded state tensors are new storage
            steps = {p: int(self.state[p]["step"]) for p in g["params"] if p in self.state and "step" in self.state[p]}
            g["_cp_step"] = max(steps.values()) if steps else 0
            for p, t in steps.items():
                self._step0[p] = g["_cp_step"] - t


class Adam(_MultiTensorOptimizer):
    """torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) -- amsgrad / maximize are not offered"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        self._step0 = {}
        # the fixed-value keys keep a saved state_dict loadable by torch.optim.Adam (and the other way round)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None))

    def _init_state(self, p, st):
        st["step"] = torch.zeros((), dtype=torch.float32)
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    @staticmethod
    def _fill_state_ptrs(it, st):
        it.m, it.v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _abi.load()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            tab = self._table(group, params)
            t = group["_cp_step"] = group.get("_cp_step", 0) + 1
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream(params[0].device).cuda_stream
            _abi.check(lib.cp_adam_multi(stream, tab["raw"].data_ptr(), tab["prefix"].data_ptr(), tab["n"], tab["blocks"], float(group["lr"]),
                                         float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), t), "cp_adam_multi")
            torch._C._increment_version(params)        # written through raw pointers: tell autograd / the eval staleness check
        return loss


class SGD(_MultiTensorOptimizer):
    """torch.optim.SGD(params, lr, momentum=0, weight_decay=0) -- dampening 0, no nesterov (what the reference uses)"""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        self._step0 = {}
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False, maximize=False,
                                      foreach=None, differentiable=False, fused=None))

    def _init_state(self, p, st):
        st["step"] = torch.zeros((), dtype=torch.float32)
        st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    @staticmethod
    def _fill_state_ptrs(it, st):
        it.m, it.v = st["momentum_buffer"].data_ptr(), None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _abi.load()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            tab = self._table(group, params)
            t = group["_cp_step"] = group.get("_cp_step", 0) + 1
            stream = torch.cuda.current_stream(params[0].device).cuda_stream
            _abi.check(lib.cp_sgd_multi(stream, tab["raw"].data_ptr(), tab["prefix"].data_ptr(), tab["n"], tab["blocks"], float(group["lr"]),
                                        float(group["momentum"]), float(group["weight_decay"]), 0), "cp_sgd_multi")
            torch._C._increment_version(params)
        return loss
