"""Backward plan of the Pix2Pix generator on the HIP kernels (autograd of networks.py:514-617).

Per level k the BatchNorm output feeds two consumers with different activations on the same pre-activation:
the next down-conv through LeakyReLU(0.2) and (through the in-place-aliased skip + the parent's uprelu) the
up-conv through ReLU; gs_bn_act_bwd_* takes both gradient sources (dz_a/act, dz_b/act_b) in one pass.
Every merged transposed conv yields dWm (per sub-pixel class, MFMA wgrad), split into dW4/dW6/dW8 and the
three architecture dot products (gs_upconv_split_wgrad); its data gradient is a stride-2 64-tap conv of dY."""
from __future__ import annotations

import math
import os
from typing import Dict

import torch

from .. import ops
from .._lib import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH
from .pix2pix_engine import cached_geom, g_tapw_identity

TAPS64 = tuple((ky - 3, kx - 3) for ky in range(8) for kx in range(8))


# the one-channel image layer's weight gradient on its direct kernel (GSSEG_IMAGE_WGRAD=0: the generic sub-pixel-class engine)
IMAGE_WGRAD_DIRECT = os.environ.get("GSSEG_IMAGE_WGRAD", "1") != "0"


def generator_backward(engine, ctx, arch, dout, need_dx):
    tdt = engine.tdt
    N, D, c, hs, ws, parts = ctx["N"], ctx["D"], ctx["c"], ctx["hs"], ctx["ws"], ctx["parts"]
    R, L, levels, ups = ctx["R"], ctx["L"], ctx["levels"], ctx["ups"]
    dev = dout.device
    H, W = hs[0], ws[0]
    S = float(2 ** round(math.log2(N * H * W)))
    inv_s = 1.0 / S
    # (data parallel: every parameter gradient is announced to the reducer as soon as it is final -- parallel.GradEmitter)
    from ..parallel import GradEmitter
    emitter = GradEmitter(getattr(engine, "grad_ready_hook", None))
    grads = emitter.grads
    names = {id(p): n for n, p in zip(engine.param_names(), engine.param_list())}
    darch = torch.zeros(arch.shape, dtype=torch.float32, device=dev)

    def empty(*shape, dtype=tdt):
        return torch.empty(shape, dtype=dtype, device=dev)

    def emit(p, g):
        emitter.emit(names[id(p)], g)

    def bn_bwd(y, coef, stats, bnmod, dz_a, sa, ca, act, dz_b=None, act_b=ACT_NONE, keep=None, kscale=1.0):
        """gradient w.r.t. the raw conv output y of  act(bn(y)) [two consumers / dropout]; emits dgamma/dbeta."""
        _, h, w, C = y.shape
        dy = empty(N, h, w, C)
        if bnmod is None:
            ops.bn_act_bwd_apply(y, dz_a, sa, ca, None, None, None, None, None, None, None, act, False, dy,
                                 dz_b=dz_b, act_b=act_b, keep_mask=keep, keep_scale=kscale)
            return dy
        nt = ops.bn_bwd_tiles_used(N, h, w, False)
        part = empty(ops.bn_partials_numel(ops.bn_bwd_tiles(N, h, w), C), dtype=torch.float32)
        ops.bn_act_bwd_reduce(y, dz_a, sa, ca, None, coef[0], coef[1], coef[2], coef[3], act, part,
                              dz_b=dz_b, act_b=act_b, keep_mask=keep, keep_scale=kscale)
        dgamma = empty(C, dtype=torch.float32)
        dbeta = empty(C, dtype=torch.float32)
        c12 = empty(2, C, dtype=torch.float32)
        ops.bn_bwd_coeffs(part, nt, C, N * h * w, inv_s, dgamma, dbeta, c12[0], c12[1])
        if not stats:
            c12.zero_()
        ops.bn_act_bwd_apply(y, dz_a, sa, ca, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], act, True, dy,
                             dz_b=dz_b, act_b=act_b, keep_mask=keep, keep_scale=kscale)
        emit(bnmod.weight, dgamma)
        emit(bnmod.bias, dbeta)
        return dy

    def upconv_bwd(d, du, cpad):
        """du: gradient w.r.t. the merged transposed conv output [N,2h,2w,cpad]; returns dR[d+1]."""
        info = ups[d]
        cell = parts[d][2]
        cin_t, cout_t, li, sm = info["cin"], info["cout"], info["li"], info["sm"]
        h, w = hs[d + 1], ws[d + 1]
        w4, w6, w8 = (cell._ops._ops[j].op.weight for j in range(3))
        xin = R[d + 1]
        if (d == 0 and IMAGE_WGRAD_DIRECT and ops.upconv8_image_wgrad_ok(cin_t, cout_t) and xin.is_contiguous()
                and xin.shape[3] == cin_t):
            # the one-channel image layer: a GEMM with ONE output column -- the direct kernel (csrc/pix2pix.hip) instead of the
            # engine's 8-cout-padded sub-pixel classes (1.7 ms at batch 32, 0.13 ms at batch 2)
            nparts, wsl, direct = 1, None, False
            dwm = torch.empty((4, 16, 1, cin_t), dtype=torch.float32, device=dev)
            ops.upconv8_image_wgrad(xin, du, dwm, N, h, w, cin_t)
        else:
            # weight gradient of the merged kernel, one sub-pixel class at a time
            geoms = []
            for cls in range(4):
                geoms.append(cached_geom(ops.geom_convT_class, N, h, w, cin_t, cpad, 8, 3, cls >> 1, cls & 1, identity_slots=True))
            # the four classes in ONE launch.  Few-pixel layers do not split K: every element is written exactly once, straight
            # into dwm; otherwise the K parts go to slabs and one ordered sum over all classes follows (deterministic, no atomics)
            nparts = ops.conv_wgrad_parts(geoms[0])
            per = 4 * 16 * cpad * cin_t
            wsl = empty(nparts * per, dtype=torch.float32) if nparts > 1 else None
            # K-split layers: the split pass below sums the slabs itself (no separate ordered-sum pass over the merged gradient)
            direct = nparts > 1 and cpad == cout_t and ops.upconv_split_wgrad_parts_ok(cin_t, cout_t)
            dwm = None if direct else torch.empty((4, 16, cpad, cin_t), dtype=torch.float32, device=dev)
            ops.conv_wgrad_det_batch(geoms, xin, du, wsl, dwm, reduce=not direct)
            if not direct and cpad != cout_t:
                dwm = dwm[:, :, :cout_t, :].contiguous()
        dw4 = torch.empty_like(w4, memory_format=torch.contiguous_format)
        dw6 = torch.empty_like(w6, memory_format=torch.contiguous_format)
        dw8 = torch.empty_like(w8, memory_format=torch.contiguous_format)
        dots = torch.zeros(3, dtype=torch.float32, device=dev)
        ops.upconv_split_wgrad(wsl if direct else dwm, w4.detach().contiguous(), w6.detach().contiguous(),
                               w8.detach().contiguous(), sm, inv_s, dw4, dw6, dw8, dots, nparts=nparts if direct else 1)
        emit(w4, dw4); emit(w6, dw6); emit(w8, dw8)
        b0 = cell._ops._ops[0].op.bias
        if b0 is not None:                                     # merged bias = sum_j sm_j b_j
            colws = empty(1024 * cpad, dtype=torch.float32)
            dbm = empty(cpad, dtype=torch.float32)
            ops.colsum(du, cpad, 0, N, 2 * h, 2 * w, 0, 0, 2 * h, 2 * w, cpad, inv_s, colws, dbm)
            dbm = dbm[:cout_t]
            for j in range(3):
                bj = cell._ops._ops[j].op.bias
                emit(bj, (sm[j] * dbm).contiguous())
                dots[j] += (dbm * bj.detach()).sum()
        # softmax backward for the architecture row (3 values)
        darch[li] = sm * (dots - (sm * dots).sum())
        # data gradient: stride-2 / pad-3 conv of du with the un-flipped merged kernel
        pd = engine._merge_pack(d, w4, w6, w8, sm, cin_t, cout_t, cpad, dgrad=True, arch=arch)
        dr = empty(N, h, w, cin_t)
        g = cached_geom(ops.make_geom, N, 2 * h, 2 * w, cpad, h, w, cin_t, h, w, TAPS64, isy=2, isx=2)
        ops.conv_igemm(g, du, pd, dr)
        return dr

    # ---- outermost: out = tanh(u0) ---------------------------------------------------------------
    info0 = ups[0]
    cpad, cout0 = info0["cpad"], info0["cout"]
    dt = torch.zeros((N, H, W, cpad), dtype=tdt, device=dev)
    ops.nchw_to_nhwc((dout.contiguous().float() * S).contiguous(), dt, cpad, 0)
    du = empty(N, H, W, cpad)
    ops.bn_act_bwd_apply(info0["u"], dt, cpad, 0, None, None, None, None, None, None, None, ACT_TANH, False, du)
    dR = upconv_bwd(0, du, cpad)                                  # gradient w.r.t. R[1]  [N,h1,w1,2c1]

    # ---- walk inwards through the up path ----------------------------------------------------------
    dR_levels = {1: dR}
    for d in range(1, D):
        info = ups[d]
        upnorm = parts[d][3]
        src = dR_levels[d]                                        # [N,h_d,w_d,2c_d]; second half belongs to block d
        du = bn_bwd(info["u"], info["coef"], info["stats"], upnorm, src, 2 * c[d], c[d], ACT_RELU,
                    keep=info["keep"], kscale=info["kscale"])
        dR_levels[d + 1] = upconv_bwd(d, du, info["cout"])

    # ---- down path, innermost first ------------------------------------------------------------------
    dL = None                                                      # gradient w.r.t. L[k] (dense), from conv k+1
    dx = None
    for k in range(D, 0, -1):
        lv = levels[k]
        conv, downnorm = parts[k - 1][0], parts[k - 1][1]
        src = dR_levels[k]
        if k == D:                                                 # innermost: only the ReLU consumer
            dy = bn_bwd(lv["y"], lv["coef"], lv["stats"], downnorm, src, c[k], 0, ACT_RELU)
        else:
            dy = bn_bwd(lv["y"], lv["coef"], lv["stats"], downnorm, src, 2 * c[k], 0, ACT_RELU,
                        dz_b=dL, act_b=ACT_LEAKY02)
        if k == 1:
            dw = torch.zeros_like(conv.weight, memory_format=torch.contiguous_format)
            ops.conv_smallcin_wgrad(ctx["x"], dy, dw, 4, 2, 1, inv_s)
            emit(conv.weight, dw)
            if need_dx:
                dx = torch.empty_like(ctx["x"])
                ops.conv_smallcin_dgrad(dy, conv.weight.detach().contiguous(), dx, 4, 2, 1, inv_s)
        else:
            cin, cout = c[k - 1], c[k]
            # K parts in slabs + ordered reduction fused with scale / unpack (one part: a plain store): deterministic
            dw = torch.empty_like(conv.weight, memory_format=torch.contiguous_format)
            wsl = empty(ops.conv_wgrad_ws_floats(lv["geom"]), dtype=torch.float32)
            ops.conv_wgrad_det(lv["geom"], lv["inp"], dy, wsl, dw, cout, cin, 16, inv_s)
            emit(conv.weight, dw)
            dL = empty(N, hs[k - 1], ws[k - 1], cin)
            gds = [cached_geom(ops.geom_conv_s2_dgrad_class, N, hs[k - 1], ws[k - 1], cin, cout, 4, 1, cls >> 1, cls & 1) for cls in range(4)]
            ops.conv_igemm_batch(gds, dy, [lv["wd"]] * 4, dL)
    if getattr(engine, "after_backward", None) is not None:
        engine.after_backward()
    return grads, darch.to(arch.device), dx
