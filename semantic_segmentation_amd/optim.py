"""Fused multi-tensor optimisers on the HIP kernels of csrc/optim.hip.

    RMSprop(net.parameters(), lr=..., weight_decay=1e-8, momentum=0.9)    # train_end2end_jsrt.py:69-70
    Adam(netG.parameters(), lr=..., betas=(0.5, 0.999))                   # pix2pix_model.py:69-72

Drop-in subclasses of torch.optim.RMSprop / torch.optim.Adam: same constructor arguments, same `state_dict()`
layout (`step`, `square_avg`, `momentum_buffer` / `exp_avg`, `exp_avg_sq`), so checkpoints and LR schedulers are
interchangeable; `step()` is ONE kernel launch over every parameter of every group with torch's single-tensor
arithmetic.  Unsupported flags of the torch classes (centered, amsgrad, maximize, sparse gradients) raise.
Parameters must be fp32 CUDA tensors: there is no CPU path.
"""
from __future__ import annotations

import math
import torch

from . import _lib
from .ops import _stream


def _h2d(values, dtype, dev):
    """small host table -> device without draining the stream: a pageable-memory `.to(device)` waits for everything queued on the
    stream (one full pipeline drain per optimiser step); pinned + non_blocking is an ordinary stream-ordered copy.  Returns
    (device tensor, pinned source): the caller keeps BOTH alive until the next step."""
    host = torch.tensor(values, dtype=dtype).pin_memory()
    return host.to(dev, non_blocking=True), host


class _FusedMixin:
    """Builds (and caches while the pointers stay the same) the device tables of one parameter group."""

    def _check_param(self, p):
        if not p.is_cuda:
            raise RuntimeError("semantic_segmentation_amd.optim runs on the MI355X only (no CPU path)")
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise TypeError("fused optimisers need contiguous fp32 parameters")
        if p.grad.is_sparse:
            raise RuntimeError("fused optimisers do not support sparse gradients")
        if p.grad.dtype != torch.float32:
            raise TypeError("fused optimisers need fp32 gradients")

    def _tables(self, group, params, grads, s1, s2):
        """Device tables of one launch.  The chunk map depends only on the tensor sizes (cached); the four pointer
        tables are rebuilt -- ONE host-to-device copy -- whenever a pointer changed (fresh .grad tensors)."""
        cache = group.setdefault("_gs_cache", {})
        dev = params[0].device
        sizes = tuple(p.numel() for p in params)
        if cache.get("sizes") != sizes:
            chunk = _lib.load().gs_optim_chunk_elems()
            ct, cs = [], []
            for i, n in enumerate(sizes):
                for start in range(0, n, chunk):
                    ct.append(i)
                    cs.append(start)
            cache["sizes"] = sizes
            cache["n"] = torch.tensor(sizes, dtype=torch.int64).to(dev)
            cache["ct"] = torch.tensor(ct, dtype=torch.int32).to(dev)
            cache["cs"] = torch.tensor(cs, dtype=torch.int64).to(dev)
            cache["nchunks"] = len(ct)
            cache["key"] = None
        key = tuple(t.data_ptr() for t in params) + tuple(t.data_ptr() for t in grads) + \
            tuple(0 if t is None else t.data_ptr() for t in s1) + tuple(0 if t is None else t.data_ptr() for t in s2)
        if cache.get("key") != key:
            T = len(params)
            cache["ptrs"], cache["ptrs_host"] = _h2d(list(key), torch.int64, dev)   # [4][T]: p, g, s1, s2
            cache["key"] = key
            cache["T"] = T
        cache["keep"] = (params, grads, s1, s2)         # the launch is asynchronous: keep every operand alive
        base, T = cache["ptrs"].data_ptr(), cache["T"]
        return dict(p=base, g=base + 8 * T, s1=base + 16 * T, s2=base + 24 * T, n=cache["n"].data_ptr(),
                    ct=cache["ct"].data_ptr(), cs=cache["cs"].data_ptr(), nchunks=cache["nchunks"])

    @staticmethod
    def _mark_updated(params):
        """The kernel wrote the parameters through raw pointers: bump their version counters, which is what the engines'
        16-bit weight-pack caches (and autograd's saved-tensor checks) key on."""
        torch.autograd.graph.increment_version(params)

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_gs_cache", None)
        return sd


class RMSprop(_FusedMixin, torch.optim.RMSprop):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False,
                 foreach=None, maximize=False, **kw):
        if centered or maximize:
            raise NotImplementedError("fused RMSprop: centered / maximize are not implemented")
        super().__init__(params, lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=momentum,
                         centered=False, foreach=False)
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, sq, buf = [], [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._check_param(p)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0) if not group.get("capturable", False) else torch.zeros((), device=p.device)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["momentum"] > 0:
                        st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                params.append(p)
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                sq.append(st["square_avg"])
                buf.append(st.get("momentum_buffer"))
            if not params:
                continue
            tb = self._tables(group, params, grads, sq, buf)
            _lib.call("gs_optim_rmsprop", tb["p"], tb["g"], tb["s1"], tb["s2"], tb["n"], tb["ct"], tb["cs"],
                      tb["nchunks"], float(group["lr"]),
                      float(group["alpha"]), float(group["eps"]), float(group["weight_decay"]), float(group["momentum"]),
                      float(self.grad_scale), _stream())
            self._mark_updated(params)
        return loss


class Adam(_FusedMixin, torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 foreach=None, maximize=False, **kw):
        if amsgrad or maximize:
            raise NotImplementedError("fused Adam: amsgrad / maximize are not implemented")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False)
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, m, v, scal = [], [], [], [], []
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._check_param(p)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                k = float(st["step"])
                scal += [group["lr"] / (1.0 - b1 ** k), math.sqrt(1.0 - b2 ** k)]
                params.append(p)
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                m.append(st["exp_avg"])
                v.append(st["exp_avg_sq"])
            if not params:
                continue
            tb = self._tables(group, params, grads, m, v)
            sc, sc_host = _h2d(scal, torch.float32, params[0].device)
            _lib.call("gs_optim_adam", tb["p"], tb["g"], tb["s1"], tb["s2"], tb["n"], tb["ct"], tb["cs"],
                      tb["nchunks"], sc.data_ptr(),
                      float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), float(self.grad_scale),
                      _stream())
            group["_gs_keepalive"] = (sc, sc_host)      # the copy and the launch are asynchronous: alive until the next step
            self._mark_updated(params)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_gs_keepalive", None)
        return sd
