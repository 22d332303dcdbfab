"""Whole-network plans of the Pix2Pix generator and PatchGAN discriminator on the HIP kernels.

Generator (reference models_pix2pix/networks.py:514-617, 8 levels at 256x256):
  * every 4x4/stride-2 down-conv is one MFMA implicit GEMM (the 1-channel outermost one a direct kernel that
    reads the fp32 NCHW mask); BatchNorm statistics come out of the conv epilogue.
  * the LeakyReLU(0.2, inplace=True) that opens every inner block mutates its input, which is also the skip
    operand of `torch.cat([x, model(x)], 1)` (:617): the skip carries leaky(x).  After the parent's in-place
    `uprelu` the up-conv therefore sees relu(leaky(x)) = relu(x): one BN pass per level writes leaky(bn(y))
    (next down-conv input) and relu(bn(y)) straight into the first half of the level's concat buffer R_k.
  * every mixed transposed conv (softmax(arch)-weighted k4/k6/k8, :486-511) runs as ONE merged k8/s2/p3
    transposed conv (exact; csrc/pix2pix.hip), split into 4 sub-pixel classes of 16 taps on the MFMA engine;
    BN (+dropout keep-mask) + ReLU write into the second half of R_{k-1}.
Discriminator (:620-665): direct 2->64 conv, three MFMA convs with BN+LeakyReLU, direct 512->1 head (fp32 logits).
Gradients are carried multiplied by a power-of-two scale (see unet_engine.py)."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from .. import ops
from .._lib import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH

_TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _dtype_of(name):
    import os
    name = name or os.environ.get("GSSEG_DTYPE", "f16")
    if name not in _TORCH_DT:
        raise ValueError("compute dtype must be 'f16' or 'bf16'")
    return name, _TORCH_DT[name]


def _need_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("semantic_segmentation_amd Pix2Pix networks run on the MI355X only (no CPU / ATen fallback)")


def _bn_coeffs(bn, partials, ntiles, C, count, training, dev, nbt_pending=None):
    """scale/shift/mean/invstd [4,C] from conv-epilogue partial sums (train) or running statistics (eval).
    nbt_pending: list collecting the `num_batches_tracked` counters of this pass -- the caller increments them with ONE
    foreach launch (_flush_nbt) instead of a 5-us kernel per BatchNorm layer."""
    coef = torch.empty((4, C), dtype=torch.float32, device=dev)
    batch_stats = training or bn.running_mean is None
    if batch_stats:
        mom = bn.momentum
        if training and bn.num_batches_tracked is not None:
            if nbt_pending is not None and mom is not None:
                nbt_pending.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        if mom is None:
            mom = 1.0 / float(bn.num_batches_tracked.item())
        upd = training and bn.running_mean is not None
        ops.bn_finalize(partials, ntiles, C, count, bn.weight.detach(), bn.bias.detach(),
                        bn.running_mean if upd else None, bn.running_var if upd else None, mom, bn.eps,
                        coef[0], coef[1], coef[2], coef[3])
    else:
        ops.bn_eval_coeffs(C, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                           coef[0], coef[1], coef[2], coef[3])
    return coef, batch_stats


# GSSEG_PIX2PIX_DIRECT_IMAGE=0: run the outermost generator layer on the MFMA engine (4 class launches + tanh + layout pass)
DIRECT_IMAGE_LAYER = os.environ.get("GSSEG_PIX2PIX_DIRECT_IMAGE", "1") != "0"


_GEOMS = {}


def cached_geom(builder, *args, identity_slots=False, **kw):
    """ops.geom_* / ops.make_geom results, memoised: building one fills 64-entry tap tables from Python (5-10 us) and a step asked for
    ~180 of them.  The geometries are read-only here (identity_slots: tap t lives in weight slot t, set once)."""
    key = (builder.__name__, args, tuple(sorted(kw.items())), identity_slots)
    g = _GEOMS.get(key)
    if g is None:
        g = builder(*args, **kw)
        if identity_slots:
            g_tapw_identity(g)
        if len(_GEOMS) > 4096:
            _GEOMS.clear()
        _GEOMS[key] = g
    return g


class _ParamIndex:
    """(name, owner module, leaf name) of every parameter in registration order -- the order autograd sees them.  Walking the module
    tree costs ~0.1-0.2 ms per call and a step asked for it a dozen times; the tree of these networks is static while the
    Parameter objects may be swapped (`.to()`), so the OWNERS are cached and the tensors are read from them on every call."""

    def _pidx(self):
        idx = self.__dict__.get("_pidx_cache")
        if idx is None:
            mods = dict(self.net.named_modules())
            idx = []
            for name, _ in self.net.named_parameters():
                head, _, leaf = name.rpartition(".")
                idx.append((name, mods[head], leaf))
            self.__dict__["_pidx_cache"] = idx
        return idx

    def param_list(self):
        return [m._parameters[leaf] for _, m, leaf in self._pidx()]

    def param_names(self):
        return [n for n, _, _ in self._pidx()]


def _flush_nbt(pending):
    if pending:
        torch._foreach_add_(pending, 1)
        pending.clear()


class _PackCache:
    """version-keyed 16-bit weight packs.  Entries built while a stream capture is running (harness hip_graphs) live in the
    graph's memory pool and are rebuilt by every replay of that graph; a capture reuses ONLY such entries (a pack built by eager
    code belongs to eager code, which may free it: a graph must not read it), eager code reuses any entry whose version matches
    (eager updates bump the versions, so it never sees a captured entry of older weights)."""

    def __init__(self):
        self._d = {}
        self._captured = []          # (key, tensor, refill) of every entry built during a stream capture: graphs hold their pointers

    def get(self, key, versions, build, refill=None):
        """refill(val): re-runs the build INTO the existing tensor `val` (kept for captured entries: refresh() below)"""
        capturing = torch.cuda.is_current_stream_capturing()
        ent = self._d.get(key)
        if ent is not None and ent[0] == versions and (ent[2] or not capturing):
            return ent[1]
        val = build()
        self._d[key] = (versions, val, capturing)
        if capturing and refill is not None:
            self._captured.append((key, val, refill))
        return val

    def refresh(self, match, versions_of):
        """Re-fill, in place, EVERY captured tensor whose key satisfies match(key).  A captured pack is written only by replays of
        the graph that built it; a graph captured LATER that merely reads it (harness: the Generator graph reads the forward packs
        the Discriminator graph merges) sees stale contents when an eager update of an input (the arch step) falls between that
        writer's replay and its own -- the eager code that made the update calls this to bring the buffers up to date.  It does not
        go by the cache entries (an eager forward in between may have replaced them): every buffer a graph may read is re-filled.
        versions_of(key) -> the current version tuple, stamped on the cache entry that still points at a re-filled tensor."""
        n = 0
        for key, val, refill in self._captured:
            if not match(key):
                continue
            refill(val)
            ent = self._d.get(key)
            if ent is not None and ent[1] is val:
                self._d[key] = (versions_of(key), val, ent[2])
            n += 1
        return n

    def clear(self):
        self._d.clear()              # (the captured tensors stay registered: graphs that read them may still be replayed)


_NOCACHE = [0]


def _ver(*ts):
    from ..unet import unet_engine
    if not unet_engine.PACK_CACHE:                   # GSSEG_PACK_CACHE=0 (loops that write parameters through `.data`)
        _NOCACHE[0] += 1
        return (_NOCACHE[0],)
    return tuple((t.data_ptr(), t._version) for t in ts)


class _EmitDict:
    """`grads[name] = g` of a backward pass routed through parallel.GradEmitter (each name is assigned once)"""

    def __init__(self, emitter):
        self._e = emitter

    def __setitem__(self, name, g):
        self._e.emit(name, g)


# =====================================================================================================
# Generator
# =====================================================================================================
class GeneratorEngine(_ParamIndex):
    def __init__(self, net, dtype=None):
        self.net = net
        self.dtype, self.tdt = _dtype_of(dtype)
        self.packs = _PackCache()
        self._layer_of, self._merge_inputs = {}, {}      # per block depth: arch row / the three kernels of its merged pack
        self.trust_versions = False      # see unet_engine.pack_reuse_allowed
        self.grad_ready_hook = self.after_backward = self.grad_fetch = None      # parallel.GradReducer.attach

    def blocks(self):
        """UnetSkipConnectionBlocks from the outermost (depth 0) to the innermost."""
        out, b = [], self.net.model
        while b is not None:
            out.append(b)
            sub = None
            for m in b.model:
                if m.__class__.__name__ == "UnetSkipConnectionBlock":
                    sub = m
            b = sub
        return out

    @staticmethod
    def _parts(block):
        conv = cell = None
        norms, drop = [], None
        for m in block.model:
            n = m.__class__.__name__
            if n == "Conv2d":
                conv = m
            elif n == "Cell_upconv":
                cell = m
            elif n == "BatchNorm2d":
                norms.append(m)
            elif n == "Dropout":
                drop = m
        if block.outermost:
            downnorm = upnorm = None
        elif block.innermost:
            downnorm, upnorm = None, norms[0]
        else:
            downnorm, upnorm = norms[0], norms[1]
        return conv, downnorm, cell, upnorm, drop

    def run(self, x, dropout_masks=None):
        from . import networks
        arch = networks.upconv_arch
        params = self.param_list()
        need_grad = torch.is_grad_enabled() and (x.requires_grad or arch.requires_grad or
                                                 any(p.requires_grad for p in params))
        return _GeneratorFunction.apply(self, self.net.training, need_grad, dropout_masks, x, arch, *params)

    # -------------------------------------------------------------------------------------------------
    def forward(self, x, arch, training, need_grad, dropout_masks):
        _need_cuda(x)
        net, tdt = self.net, self.tdt
        blocks = self.blocks()
        D = len(blocks)
        N, cin0, H, W = x.shape
        if cin0 > 4:
            raise NotImplementedError("generator input_nc above 4 is not supported by the direct first-layer kernel")
        if H % (1 << D) or W % (1 << D):
            raise ValueError(f"input size must be a multiple of {1 << D}")
        dev = x.device
        x = x.contiguous().float()
        arch = arch.to(dev)
        from ..unet.unet_engine import pack_reuse_allowed
        if not pack_reuse_allowed(need_grad, self.trust_versions):
            self.packs.clear()       # training forwards always re-pack: `.data` writes (Betty) bump no version counter

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        parts = [self._parts(b) for b in blocks]
        c = [cin0] + [p[0].out_channels for p in parts]            # c[k]: channels after k down-convs
        hs = [H >> k for k in range(D + 1)]
        ws = [W >> k for k in range(D + 1)]
        ctx = dict(N=N, D=D, c=c, hs=hs, ws=ws, parts=parts, training=training, x=x, levels=[None] * (D + 1),
                   ups=[None] * D)
        L: List[Optional[torch.Tensor]] = [None] * (D + 1)          # leaky(bn(y_k)), dense: next down-conv input
        R: List[Optional[torch.Tensor]] = [None] * (D + 1)          # [N,h_k,w_k,2c_k]: relu(skip) | relu(up)

        nbt_pending = []                 # num_batches_tracked counters of this pass: ONE foreach increment at the end
        # ---- down path -------------------------------------------------------------------------------
        conv0 = parts[0][0]
        y1 = empty(N, hs[1], ws[1], c[1])
        ops.conv_smallcin_fwd(x, conv0.weight.detach().contiguous(), None, y1, None, 4, 2, 1)
        L[1] = empty(N, hs[1], ws[1], c[1])
        R[1] = empty(N, hs[1], ws[1], 2 * c[1])
        ops.bn_act_apply(y1, None, None, ACT_LEAKY02, L[1], c[1], 0)
        ops.bn_act_apply(y1, None, None, ACT_RELU, R[1], 2 * c[1], 0)
        ctx["levels"][1] = dict(y=y1, coef=None, stats=False)
        for k in range(2, D + 1):
            conv, downnorm = parts[k - 1][0], parts[k - 1][1]
            wf, wd = self.packs.get(("down", k), _ver(conv.weight),
                                    lambda conv=conv: self._pack_conv(conv.weight))
            g = cached_geom(ops.geom_conv, N, hs[k - 1], ws[k - 1], c[k - 1], c[k], 4, 2, 1)
            y = empty(N, hs[k], ws[k], c[k])
            has_bn = downnorm is not None
            mt = ops.conv_igemm_mtiles(g)
            use_stats = has_bn and (training or downnorm.running_mean is None)
            part = empty(ops.bn_partials_numel(mt, c[k]), dtype=torch.float32) if use_stats else None
            ops.conv_igemm(g, L[k - 1], wf, y, None, part)
            coef, stats = (None, False)
            if has_bn:
                coef, stats = _bn_coeffs(downnorm, part, mt, c[k], N * hs[k] * ws[k], training, dev, nbt_pending)
            sc, sh = (coef[0], coef[1]) if has_bn else (None, None)
            ctx["levels"][k] = dict(y=y, coef=coef, stats=stats, geom=g, wd=wd, inp=L[k - 1])
            if k < D:
                L[k] = empty(N, hs[k], ws[k], c[k])
                R[k] = empty(N, hs[k], ws[k], 2 * c[k])
                ops.bn_act_apply(y, sc, sh, ACT_LEAKY02, L[k], c[k], 0)
                ops.bn_act_apply(y, sc, sh, ACT_RELU, R[k], 2 * c[k], 0)
            else:                                                    # innermost: conv -> uprelu -> upconv
                R[k] = empty(N, hs[k], ws[k], c[k])
                ops.bn_act_apply(y, sc, sh, ACT_RELU, R[k], c[k], 0)

        # ---- up path -----------------------------------------------------------------------------------
        sm_all = torch.softmax(arch.detach().float(), dim=-1).contiguous()      # every layer's mixing weights in one launch
        out = None
        mask_i = 0
        for d in range(D - 1, -1, -1):                               # block depth d consumes R[d+1]
            conv, downnorm, cell, upnorm, drop = parts[d]
            cin_t = R[d + 1].shape[3]
            cout_t = cell._ops._ops[0].op.weight.shape[1]
            li = cell._layer_index
            w4, w6, w8 = (cell._ops._ops[j].op.weight for j in range(3))
            sm = sm_all[li]
            cpad = cout_t if cout_t % 8 == 0 else ((cout_t + 7) // 8) * 8
            pf = self._merge_pack(d, w4, w6, w8, sm, cin_t, cout_t, cpad, dgrad=False, arch=arch, li=li)
            h, w = hs[d + 1], ws[d + 1]
            H2, W2 = 2 * h, 2 * w
            bias = None
            if cell._ops._ops[0].op.bias is not None:
                bias = sum(sm[j] * cell._ops._ops[j].op.bias.detach() for j in range(3)).float().contiguous()
            if d > 0:
                u = empty(N, H2, W2, cout_t)
                mt = ops.conv_igemm_mtiles(cached_geom(ops.geom_convT_class, N, h, w, cin_t, cout_t, 8, 3, 0, 0, identity_slots=True))
                use_stats = training or upnorm.running_mean is None
                part = empty(ops.bn_partials_numel(4 * mt, cout_t), dtype=torch.float32) if use_stats else None
                geoms = []
                for cls in range(4):
                    geoms.append(cached_geom(ops.geom_convT_class, N, h, w, cin_t, cout_t, 8, 3, cls >> 1, cls & 1, identity_slots=True))
                pslices = [part[cls * mt * 2 * cout_t:] for cls in range(4)] if part is not None else None
                # the four sub-pixel classes in ONE launch (each is latency bound on its own at the script's batch size)
                ops.conv_igemm_batch(geoms, R[d + 1], [pf[cls] for cls in range(4)], u, bias, pslices)
                coef, stats = _bn_coeffs(upnorm, part, 4 * mt, cout_t, N * H2 * W2, training, dev, nbt_pending)
                keep, kscale = None, 1.0
                if drop is not None and training and drop.p > 0:
                    if dropout_masks is not None:
                        keep = dropout_masks[mask_i].to(dev).contiguous()
                    else:
                        keep = (torch.rand((N, H2, W2, cout_t), device=dev) >= drop.p).to(torch.uint8)
                    kscale = 1.0 / (1.0 - drop.p)
                if drop is not None:
                    mask_i += 1
                ops.bn_act_apply(u, coef[0], coef[1], ACT_RELU, R[d], 2 * c[d], c[d], None, keep, kscale)
                ctx["ups"][d] = dict(u=u, coef=coef, stats=stats, keep=keep, kscale=kscale, sm=sm, cin=cin_t,
                                     cout=cout_t, li=li)
            else:                                                    # outermost: + bias, tanh, fp32 NCHW image
                u = empty(N, H2, W2, cpad)
                bpad = None
                if bias is not None:
                    bpad = torch.zeros(cpad, dtype=torch.float32, device=dev)
                    bpad[:cout_t] = bias
                out = torch.empty((N, cout_t, H2, W2), dtype=torch.float32, device=dev)
                if DIRECT_IMAGE_LAYER and cpad == 8 and ops.upconv8_image_fits(cin_t, cout_t):
                    # direct kernel: bias + tanh + fp32 NCHW in one pass (the MFMA engine would use 1 of 64 N columns)
                    ops.upconv8_image_fwd(R[1], pf, bpad, out, u if need_grad else None, N, h, w, cin_t, cout_t, ACT_TANH)
                else:
                    geoms = []
                    for cls in range(4):
                        geoms.append(cached_geom(ops.geom_convT_class, N, h, w, cin_t, cpad, 8, 3, cls >> 1, cls & 1, identity_slots=True))
                    ops.conv_igemm_batch(geoms, R[1], [pf[cls] for cls in range(4)], u, bpad, None)
                    t = empty(N, H2, W2, cpad)
                    ops.bn_act_apply(u, None, None, ACT_TANH, t, cpad, 0)
                    ops.nhwc_to_nchw(t, out, cpad, 0)
                ctx["ups"][0] = dict(u=u, sm=sm, cin=cin_t, cout=cout_t, cpad=cpad, li=li)
        ctx["R"], ctx["L"] = R, L
        _flush_nbt(nbt_pending)
        return out, (ctx if need_grad else None)

    def _pack_conv(self, w):
        cout, cin, kh, kw = w.shape
        wf = torch.empty((kh * kw, cout, cin), dtype=self.tdt, device=w.device)
        wd = torch.empty((kh * kw, cin, cout), dtype=self.tdt, device=w.device)
        ops.pack_weight(w.detach().contiguous(), wf, wd, False)
        return wf, wd

    def refresh_arch_packs(self, arch):
        """After an EAGER update of the architecture tensor while hipGraphs hold merged forward packs (harness.EndToEndTrainer):
        re-merge, in place, the captured packs whose version key no longer matches (see _PackCache.refresh)."""
        def versions_of(key):
            w4, w6, w8 = self._merge_inputs[key[1:]]
            return _ver(w4, w6, w8, arch)
        # (forward packs only: a graph that needs the data-gradient packs merges them itself)
        return self.packs.refresh(lambda k: k[0] == "merged" and not k[2] and k[1:] in self._merge_inputs, versions_of)

    def _merge_pack(self, d, w4, w6, w8, sm, cin_t, cout_t, cpad, dgrad, arch=None, li=None, out=None):
        """class-major forward pack [4][16][cpad][cin] (or dgrad pack [64][cin][cpad]) of the merged kernel.  Cached per
        (weights, architecture parameters) version: in the end-to-end loop the generator runs three times per iteration
        (Generator, Discriminator and Unet steps) between two updates of its weights, and re-merging reads all 1.09 GB
        of fp32 kernels each time."""
        if arch is not None:
            if li is None:
                li = self._layer_of.get(d)
            else:
                self._layer_of[d] = li
            self._merge_inputs[(d, dgrad, cpad)] = (w4, w6, w8)

            def refill(out, li=li):
                sm_now = torch.softmax(arch.detach().float(), dim=-1)[li].contiguous()
                self._merge_pack(d, w4, w6, w8, sm_now, cin_t, cout_t, cpad, dgrad, out=out)
            return self.packs.get(("merged", d, dgrad, cpad), _ver(w4, w6, w8, arch),
                                  lambda: self._merge_pack(d, w4, w6, w8, sm, cin_t, cout_t, cpad, dgrad),
                                  refill if li is not None else None)
        dev = w8.device
        if cpad == cout_t:
            w4p, w6p, w8p = w4.detach().contiguous(), w6.detach().contiguous(), w8.detach().contiguous()
        else:                                    # 1-channel image head: pad Cout to 8 with zero kernels
            def pad(w):
                z = torch.zeros((w.shape[0], cpad, w.shape[2], w.shape[3]), dtype=torch.float32, device=dev)
                z[:, :cout_t] = w.detach()
                return z
            w4p, w6p, w8p = pad(w4), pad(w6), pad(w8)
        if dgrad:
            pd = torch.empty((64, cin_t, cpad), dtype=self.tdt, device=dev) if out is None else out
            ops.upconv_merge_pack(w4p, w6p, w8p, sm, None, pd, None)
            return pd
        pf = torch.empty((4, 16, cpad, cin_t), dtype=self.tdt, device=dev) if out is None else out
        ops.upconv_merge_pack(w4p, w6p, w8p, sm, pf, None, None)
        return pf

    def invalidate_packs(self):
        self.packs.clear()


def g_tapw_identity(g):
    """class-major packs: tap t of the class lives in slot t."""
    for t in range(g.ntaps):
        g.tap_w[t] = t


class _GeneratorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, training, need_grad, dropout_masks, x, arch, *plist):
        out, ectx = engine.forward(x, arch, training, need_grad, dropout_masks)
        ctx.engine, ctx.ectx, ctx.plist, ctx.arch = engine, ectx, plist, arch
        ctx.x_needs_grad = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.ectx is None:
            raise RuntimeError("UnetGenerator forward ran without gradient tracking")
        from .pix2pix_backward import generator_backward
        grads, darch, dx = generator_backward(ctx.engine, ctx.ectx, ctx.arch, dout, ctx.x_needs_grad)
        names = ctx.engine.param_names()
        fetch = ctx.engine.grad_fetch               # data parallel: the reduced gradients
        out = [(fetch(n) if (fetch is not None and n in grads) else grads.get(n)) if p.requires_grad else None
               for n, p in zip(names, ctx.plist)]
        return (None, None, None, None, dx, darch, *out)


# =====================================================================================================
# Discriminator
# =====================================================================================================
class DiscriminatorEngine(_ParamIndex):
    def __init__(self, net, dtype=None):
        self.net = net
        self.dtype, self.tdt = _dtype_of(dtype)
        self.packs = _PackCache()
        self.trust_versions = False      # see unet_engine.pack_reuse_allowed
        self.grad_ready_hook = self.after_backward = self.grad_fetch = None      # parallel.GradReducer.attach

    def run(self, x):
        params = self.param_list()
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _DiscriminatorFunction.apply(self, self.net.training, need_grad, x, *params)

    def layout(self):
        seq = list(self.net.model)
        convs = [(i, m) for i, m in enumerate(seq) if isinstance(m, torch.nn.Conv2d)]
        stages = []
        for idx, (i, conv) in enumerate(convs):
            bn = seq[i + 1] if i + 1 < len(seq) and isinstance(seq[i + 1], torch.nn.BatchNorm2d) else None
            stages.append((i, conv, bn))
        return stages

    def forward(self, x, training, need_grad):
        _need_cuda(x)
        tdt = self.tdt
        dev = x.device
        x = x.contiguous().float()
        N, cin0, H, W = x.shape
        stages = self.layout()
        from ..unet.unet_engine import pack_reuse_allowed
        if not pack_reuse_allowed(need_grad, self.trust_versions):
            self.packs.clear()       # training forwards always re-pack: `.data` writes (Betty) bump no version counter
        if cin0 > 4:
            raise NotImplementedError("discriminator input_nc above 4 is not supported by the direct first-layer kernel")

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        recs = []
        nbt_pending = []
        # stage 0: direct conv + bias + LeakyReLU (no norm)
        i0, conv0, _ = stages[0]
        k, s, p = conv0.kernel_size[0], conv0.stride[0], conv0.padding[0]
        h, w = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
        z = empty(N, h, w, conv0.out_channels)
        ops.conv_smallcin_fwd(x, conv0.weight.detach().contiguous(),
                              conv0.bias.detach() if conv0.bias is not None else None, z, None, k, s, p, ACT_LEAKY02)
        recs.append(dict(kind="first", conv=conv0, x=x, z=z, k=k, s=s, p=p, name=f"model.{i0}"))
        cur, ch, cw, cc = z, h, w, conv0.out_channels
        for (i, conv, bn) in stages[1:-1]:
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            if conv.bias is not None or bn is None:
                raise NotImplementedError("PatchGAN middle convs are expected as conv(no bias) -> BatchNorm -> LeakyReLU")
            wf, wd = self.packs.get(("conv", i), _ver(conv.weight), lambda conv=conv: self._pack(conv.weight))
            g = cached_geom(ops.geom_conv, N, ch, cw, cc, conv.out_channels, k, s, p)
            oh, ow = g.OH, g.OW
            y = empty(N, oh, ow, conv.out_channels)
            mt = ops.conv_igemm_mtiles(g)
            use_stats = training or bn.running_mean is None
            part = empty(ops.bn_partials_numel(mt, conv.out_channels), dtype=torch.float32) if use_stats else None
            ops.conv_igemm(g, cur, wf, y, None, part)
            coef, stats = _bn_coeffs(bn, part, mt, conv.out_channels, N * oh * ow, training, dev, nbt_pending)
            z = empty(N, oh, ow, conv.out_channels)
            ops.bn_act_apply(y, coef[0], coef[1], ACT_LEAKY02, z, conv.out_channels, 0)
            recs.append(dict(kind="mid", conv=conv, bn=bn, inp=cur, y=y, coef=coef, stats=stats, geom=g, wd=wd,
                             k=k, s=s, p=p, ih=ch, iw=cw, cin=cc, name=f"model.{i}", bnname=f"model.{i + 1}"))
            cur, ch, cw, cc = z, oh, ow, conv.out_channels
        il, convl, _ = stages[-1]
        k, s, p = convl.kernel_size[0], convl.stride[0], convl.padding[0]
        oh, ow = ops.conv_out_size(ch, k, s, p), ops.conv_out_size(cw, k, s, p)
        logits = torch.empty((N, convl.out_channels, oh, ow), dtype=torch.float32, device=dev)
        ops.conv_smallcout_fwd(cur, convl.weight.detach().contiguous(),
                               convl.bias.detach() if convl.bias is not None else None, logits, k, s, p)
        recs.append(dict(kind="last", conv=convl, inp=cur, k=k, s=s, p=p, ih=ch, iw=cw, cin=cc, name=f"model.{il}"))
        _flush_nbt(nbt_pending)
        return logits, (dict(recs=recs, N=N, H=H, W=W) if need_grad else None)

    def _pack(self, w):
        cout, cin, kh, kw = w.shape
        wf = torch.empty((kh * kw, cout, cin), dtype=self.tdt, device=w.device)
        wd = torch.empty((kh * kw, cin, cout), dtype=self.tdt, device=w.device)
        ops.pack_weight(w.detach().contiguous(), wf, wd, False)
        return wf, wd

    def backward(self, ctx, dlogits, need_dx):
        tdt = self.tdt
        recs, N = ctx["recs"], ctx["N"]
        dev = dlogits.device
        S = float(2 ** round(math.log2(max(dlogits.numel(), 1))))
        inv_s = 1.0 / S
        from ..parallel import GradEmitter
        emitter = GradEmitter(self.grad_ready_hook)          # data parallel: announced as soon as final
        grads = _EmitDict(emitter)

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        last = recs[-1]
        conv = last["conv"]
        dl = dlogits.contiguous().float() * S
        dw = torch.zeros_like(conv.weight, memory_format=torch.contiguous_format)
        db = torch.zeros_like(conv.bias) if conv.bias is not None else None
        dz = empty(*last["inp"].shape)
        ops.conv_smallcout_bwd(last["inp"], conv.weight.detach().contiguous(), dl, dz, dw, db,
                               last["k"], last["s"], last["p"], inv_s)
        grads[last["name"] + ".weight"] = dw
        if db is not None:
            grads[last["name"] + ".bias"] = db
        for rec in reversed(recs[1:-1]):
            conv, coef = rec["conv"], rec["coef"]
            y = rec["y"]
            _, oh, ow, cout = y.shape
            nt = ops.bn_bwd_tiles_used(N, oh, ow, False)
            part = empty(ops.bn_partials_numel(ops.bn_bwd_tiles(N, oh, ow), cout), dtype=torch.float32)
            ops.bn_act_bwd_reduce(y, dz, cout, 0, None, coef[0], coef[1], coef[2], coef[3], ACT_LEAKY02, part)
            dgamma = empty(cout, dtype=torch.float32)
            dbeta = empty(cout, dtype=torch.float32)
            c12 = empty(2, cout, dtype=torch.float32)
            ops.bn_bwd_coeffs(part, nt, cout, N * oh * ow, inv_s, dgamma, dbeta, c12[0], c12[1])
            if not rec["stats"]:
                c12.zero_()
            dy = empty(N, oh, ow, cout)
            ops.bn_act_bwd_apply(y, dz, cout, 0, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1],
                                 ACT_LEAKY02, True, dy)
            k, s, p, cin = rec["k"], rec["s"], rec["p"], rec["cin"]
            # K parts in slabs + ordered reduction fused with scale / unpack (one part: a plain store): deterministic
            dw = torch.empty_like(conv.weight, memory_format=torch.contiguous_format)
            wsl = empty(ops.conv_wgrad_ws_floats(rec["geom"]), dtype=torch.float32)
            ops.conv_wgrad_det(rec["geom"], rec["inp"], dy, wsl, dw, cout, cin, k * k, inv_s)
            grads[rec["name"] + ".weight"] = dw
            grads[rec["bnname"] + ".weight"] = dgamma
            grads[rec["bnname"] + ".bias"] = dbeta
            dz = empty(N, rec["ih"], rec["iw"], cin)
            if s == 1:
                ops.conv_igemm(cached_geom(ops.geom_conv_dgrad_s1, N, rec["ih"], rec["iw"], cin, cout, k, p), dy, rec["wd"], dz)
            else:
                gds = [cached_geom(ops.geom_conv_s2_dgrad_class, N, rec["ih"], rec["iw"], cin, cout, k, p, cls >> 1, cls & 1)
                       for cls in range(4)]
                ops.conv_igemm_batch(gds, dy, [rec["wd"]] * 4, dz)
        first = recs[0]
        conv = first["conv"]
        z = first["z"]
        dy = empty(*z.shape)
        # z = leaky(conv + bias): the activation gradient only needs the sign, which z preserves
        ops.bn_act_bwd_apply(z, dz, z.shape[3], 0, None, None, None, None, None, None, None, ACT_LEAKY02, False, dy)
        dw = torch.zeros_like(conv.weight, memory_format=torch.contiguous_format)
        ops.conv_smallcin_wgrad(first["x"], dy, dw, first["k"], first["s"], first["p"], inv_s)
        grads[first["name"] + ".weight"] = dw
        if conv.bias is not None:
            ws = empty(1024 * z.shape[3], dtype=torch.float32)
            db = empty(z.shape[3], dtype=torch.float32)
            ops.colsum(dy, z.shape[3], 0, N, z.shape[1], z.shape[2], 0, 0, z.shape[1], z.shape[2], z.shape[3], inv_s, ws, db)
            grads[first["name"] + ".bias"] = db
        dx = None
        if need_dx:
            dx = torch.empty_like(first["x"])
            ops.conv_smallcin_dgrad(dy, conv.weight.detach().contiguous(), dx, first["k"], first["s"], first["p"], inv_s)
        if self.after_backward is not None:
            self.after_backward()
        return emitter.grads, dx


class _DiscriminatorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, training, need_grad, x, *plist):
        logits, ectx = engine.forward(x, training, need_grad)
        ctx.engine, ctx.ectx, ctx.plist = engine, ectx, plist
        ctx.x_needs_grad = x.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.ectx is None:
            raise RuntimeError("NLayerDiscriminator forward ran without gradient tracking")
        grads, dx = ctx.engine.backward(ctx.ectx, dlogits, ctx.x_needs_grad)
        names = ctx.engine.param_names()
        fetch = ctx.engine.grad_fetch               # data parallel: the reduced gradients
        out = [(fetch(n) if (fetch is not None and n in grads) else grads.get(n)) if p.requires_grad else None
               for n, p in zip(names, ctx.plist)]
        return (None, None, None, dx, *out)
