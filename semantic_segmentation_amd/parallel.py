"""Data parallelism for the hot path: one process per GPU, identical replicas, ONE exchange per step --
a sum/avg all-reduce of the parameter gradients over RCCL (torch.distributed backend "nccl" on ROCm),
bucketed in backward order and launched while the rest of backward still runs.

The reference has no collective on this path (only single-process nn.DataParallel,
unreachable from its scripts), so this is new.  Semantics follow DDP: per-rank BatchNorm statistics and
per-rank Dice; rank 0's buffers are broadcast at start.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 124 MB fp32 gradient set is ~0.2 ms as a direct
reduce-scatter+all-gather but ~1.4 ms through a single-link ring, so buckets are few and large (default
32 MB) and always overlapped with the remaining backward kernels."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

# CUs left to RCCL's kernels while the persistent conv / weight-gradient / up-conv grids run: one 160 KB-LDS block per CU holds
# every CU until its items are done, so a bucket's all-reduce kernel starts at the next kernel boundary (every ~150 us in the
# U-Net backward) rather than immediately.  GSSEG_RCCL_CUS=n makes GradReducer.attach() cap those grids at 256 - n blocks when
# world > 1.  DEFAULT 0 (no cap): measured on one GPU (tools/bench_grid_cap.py, profiles/r03_grid_cap.txt) a cap of 248 / 240 /
# 224 blocks costs +8.2 / +8.4 / +8.5 % of the bs=32 step -- the layers' item counts are multiples of 256, so ANY smaller grid
# adds a whole round of items (and doubles the launches that have exactly 256) -- which is more than the 124 MB exchange can
# cost un-overlapped over xGMI (0.2 ms direct, 1.4 ms ring = 1.6..11 %).  The knob is there for the 8-GPU measurement.
RCCL_CUS = int(os.environ.get("GSSEG_RCCL_CUS", "0"))


class GradEmitter:
    """What an engine's backward uses to hand out parameter gradients: `emit(name, g)` accumulates (a module used twice --
    the shared BatchNorm3d of the 3-D decoder blocks -- emits twice) and, once a parameter has received all `expected[name]`
    contributions, announces it to `hook` (GradReducer.ready) so that the bucket all-reduce overlaps the rest of backward."""

    def __init__(self, hook: Optional[Callable[[str, torch.Tensor], None]] = None, expected: Optional[Dict[str, int]] = None):
        self.grads: Dict[str, torch.Tensor] = {}
        self.hook = hook
        self.expected = expected or {}
        self._count: Dict[str, int] = {}

    def emit(self, name: str, g: torch.Tensor):
        c = self._count.get(name, 0) + 1
        self._count[name] = c
        self.grads[name] = g if c == 1 else self.grads[name] + g
        if self.hook is not None and c == self.expected.get(name, 1):
            self.hook(name, self.grads[name])


class GradReducer:
    """Owns flat fp32 gradient buckets; plugs into UNetEngine via grad_alloc / grad_ready_hook / after_backward.

    Parameters are laid out in REVERSE registration order (= the order backward finishes them), cut into
    buckets of ~bucket_bytes.  When the last gradient of a bucket is announced its all-reduce is issued
    asynchronously; `finish()` makes the current stream wait for all of them."""

    def __init__(self, named_params, bucket_bytes: int = 32 << 20, group=None, average: bool = True,
                 force_collective: bool = False, dtype: torch.dtype = torch.float32):
        """dtype=torch.bfloat16: the buckets (and the exchange) are bf16 -- half the xGMI bytes for the generator's 1.09 GB
        of fp32 gradients; gradients are then produced in the engine's own fp32 tensors and cast into the bucket."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("bucket dtype must be float32 or bfloat16")
        self.dtype = dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: issue the all-reduces even in a 1-rank group (exercises the RCCL path on one GPU)
        self.collective = (self.world > 1) or (force_collective and dist.is_initialized())
        items = [(n, p) for n, p in named_params if p.requires_grad][::-1]
        if not items:
            raise ValueError("no trainable parameters")
        self._params = dict(items)
        dev = items[0][1].device
        total = sum(p.numel() for _, p in items)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.views: Dict[str, torch.Tensor] = {}
        self.bucket_of: Dict[str, int] = {}
        self.buckets: List[dict] = []
        off, start, names = 0, 0, []
        for n, p in items:
            self.views[n] = self.flat[off:off + p.numel()].view(p.shape)
            names.append(n)
            off += p.numel()
            if (off - start) * self.flat.element_size() >= bucket_bytes:
                self._close(start, off, names)
                start, names = off, []
        if names:
            self._close(start, off, names)
        self.op = dist.ReduceOp.AVG if (average and dist.is_initialized() and dist.get_backend(group) == "nccl") \
            else dist.ReduceOp.SUM
        self.average = average
        self._works: list = []
        self._pending: List[int] = []
        self._live: List[int] = []          # per bucket: gradients of still-trainable parameters announced in this backward
        self.issued = 0                     # all-reduces issued by the last backward (tests read it before finish())

    def _close(self, a, b, names):
        idx = len(self.buckets)
        self.buckets.append({"range": (a, b), "names": list(names)})
        for n in names:
            self.bucket_of[n] = idx

    # -- engine hooks --------------------------------------------------------------------------------
    def begin(self):
        self._works = []
        self._pending = [len(b["names"]) for b in self.buckets]
        self._live = [0] * len(self.buckets)
        self.issued = 0

    def alloc(self, name: str, like: torch.Tensor) -> torch.Tensor:
        return self.views[name]

    def ready(self, name: str, grad: torch.Tensor):
        if not self._pending:
            self.begin()
        b = self.bucket_of.get(name)
        if b is None:                        # not trainable when the reducer was built: nothing to exchange
            return
        if self._params[name].requires_grad:
            if grad.data_ptr() != self.views[name].data_ptr():
                self.views[name].copy_(grad)
            self._live[b] += 1
        # (a parameter frozen since -- the discriminator inside the Generator problem -- still counts as announced, but a bucket
        # made of frozen parameters only is not exchanged: its result would be discarded.  Freezing must match across ranks.)
        self._pending[b] -= 1
        if self._pending[b] < 0:
            raise RuntimeError(f"GradReducer: gradient '{name}' announced more often than its bucket expects (a parameter used "
                               "twice must be announced once, after its last contribution: parallel.GradEmitter)")
        if self._pending[b] == 0 and self._live[b] > 0 and self.collective:
            a, e = self.buckets[b]["range"]
            self._works.append(dist.all_reduce(self.flat[a:e], op=self.op, group=self.group, async_op=True))
            self.issued += 1

    def finish(self):
        for w in self._works:
            w.wait()
        missing = [n for i, b in enumerate(self.buckets) if self._pending and self._pending[i] != 0 for n in b["names"]]
        self._works = []
        pending_any = bool(self._pending)
        self._pending = []
        if pending_any and missing:
            # a bucket whose gradients were not all announced was never reduced: fetch() would hand back LOCAL gradients
            raise RuntimeError("GradReducer.finish: %d bucket(s) incomplete -- some of their gradients were never announced "
                               "(first names: %s)" % (len({self.bucket_of[n] for n in missing}), ", ".join(missing[:4])))
        if self.world > 1 and self.average and self.op == dist.ReduceOp.SUM:
            self.flat.mul_(1.0 / self.world)

    def fetch(self, name: str) -> torch.Tensor:
        """the reduced gradient as a fresh fp32 tensor (the bucket is overwritten by the next backward)"""
        return self.views[name].to(torch.float32, copy=True)

    def attach(self, engine):
        """UNetEngine / UNet3DEngine / GeneratorEngine / DiscriminatorEngine: gradients are announced as they become final,
        `after_backward` waits for the collectives, autograd receives `fetch(name)`."""
        if hasattr(engine, "grad_alloc"):
            engine.grad_alloc = self.alloc if self.dtype == torch.float32 else None      # fp32 buckets: produced in place
        engine.grad_ready_hook = self.ready
        engine.after_backward = self.finish
        engine.grad_fetch = self.fetch
        if self.world > 1 and RCCL_CUS > 0 and self.flat.is_cuda:
            from . import ops
            ops.set_persistent_grid(max(8, 256 - RCCL_CUS))
        return self


def all_reduce_gradients(params, bucket_bytes: int = 64 << 20, group=None, average: bool = True) -> int:
    """Post-backward gradient exchange for modules whose engines have no bucket hooks (UNet3D, the Pix2Pix pair,
    multi-forward steps such as harness.EndToEndTrainer): the `.grad`s are packed into flat fp32 buckets in reverse
    registration order, each bucket all-reduced asynchronously (the copies of the next bucket overlap the
    collective of the previous one) and scattered back.  Parameters without gradient are skipped on EVERY rank
    alike (the set must match across ranks, as in DDP).  Returns the number of elements exchanged."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    plist = [p for p in params if p.grad is not None][::-1]
    if not plist:
        return 0
    use_avg = average and dist.get_backend(group) == "nccl"
    op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
    pending, total, i = [], 0, 0
    while i < len(plist):
        chunk, nbytes = [], 0
        while i < len(plist) and (not chunk or nbytes + plist[i].numel() * 4 <= bucket_bytes):
            chunk.append(plist[i])
            nbytes += plist[i].numel() * 4
            i += 1
        flat = torch.cat([p.grad.reshape(-1).float() for p in chunk])
        pending.append((dist.all_reduce(flat, op=op, group=group, async_op=True), flat, chunk))
        total += flat.numel()
    for work, flat, chunk in pending:
        work.wait()
        if average and not use_avg:
            flat.mul_(1.0 / world)
        off = 0
        for p in chunk:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return total


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """Replicas start identical: parameters and buffers of rank `src` (DDP broadcast_buffers semantics)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = list(module.parameters()) + list(module.buffers())
    for t in tensors:
        dist.broadcast(t.data, src=src, group=group)
    # the engines key their 16-bit weight packs / folded-BN caches on (data_ptr, _version): a write through `.data`
    # does not bump the version, so do it here -- a broadcast AFTER a forward must not leave stale packs behind
    torch.autograd.graph.increment_version(tensors)


def shard_batch(n_global: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global batch (rank r owns samples r*B/world ...)."""
    per = n_global // world
    if per * world != n_global:
        raise ValueError("global batch must be divisible by the world size")
    return rank * per, (rank + 1) * per
