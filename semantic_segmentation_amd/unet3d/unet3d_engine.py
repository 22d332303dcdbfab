"""Whole-network plan of the GenSeg-3D UNet3D (GenSeg-3D/UNet3D/unet3d.py:14-126) on the HIP kernels.

A volume is a stack of NHWC images: every tensor is [NB*D, H, W, C] 16-bit, so the 2-D BatchNorm / activation /
column-sum kernels apply unchanged (statistics over NB*D*H*W).  Conv3d 3x3x3 is the generic MFMA implicit GEMM
with 27 taps whose depth component moves the image index (GsConvGeom.tap_dz / Dg / Din); ConvTranspose3d k2 s2 is
eight one-tap sub-voxel classes written into the concat buffer (up channels FIRST, :80).  The 1-channel first
convolution runs on the direct kernel over a depth-unfolded view of the fp32 volume ([NB*D, 3, H, W]).
Bias in front of train-mode BatchNorm cancels in the output; it is folded into the running mean (train) or into
the shift (eval), and its gradient is the column sum of the conv-output gradient."""
from __future__ import annotations

import math
import os
from typing import Dict, List

import torch
import torch.nn.functional as F

from .. import ops
from .._lib import ACT_RELU

_TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}
K3 = [(kd, ky, kx) for kd in range(3) for ky in range(3) for kx in range(3)]


STAGES3D = ([f"a_block{k}.conv{c}" for k in (1, 2, 3) for c in (1, 2)] + ["bottleNeck.conv1", "bottleNeck.conv2"] +
            [s_ for k in (3, 2, 1) for s_ in (f"s_block{k}.upconv1", f"s_block{k}.conv1", f"s_block{k}.conv2")])
# Where the 16-bit error of the UNet3D logits is made (tools/parity_fp8_sim.py --net unet3d, profiles/r04_fp8_sim_unet3d_32.json): the
# two finest levels.  With K = 27 * C the rounding of the conv INPUTS (activations and weights) dominates -- pair storage alone
# leaves 2.0e-3 -- so the "mixed" plan keeps pairs everywhere and runs the correction segments on the convs of levels 0 and 1
# (a_block1.conv1 reads the fp32 volume on the direct kernel): simulated max |dlogit| 4.8e-4 (16-bit everywhere: 2.0e-3 .. 2.4e-3).
# Round 4, measured on the MI355X (tools/experiments/plan3d_sweep.py; max |dlogit| on the two 16^3 fixtures / 128^3 step): all seven convs of
# levels 0-1 4.7e-4 / 20.29 ms; without s_block2.conv1 (the level-1 decoder entry, K = 640 per tap) 5.2e-4 / 19.88 ms -- taken;
# without both s_block2 convs 6.5e-4 / 19.55 ms; without a_block2 9.2e-4; level 0 only 9.7e-4.
# BatchNorm + ReLU + MaxPool3d of an analysis block in one pass (GSSEG_POOL3D_FUSED=0: apply, then gs_maxpool3d_fwd_pair)
FUSED_POOL3D = os.environ.get("GSSEG_POOL3D_FUSED", "1") != "0"
MIXED3D_XW = ("a_block1.conv2", "a_block2.conv1", "a_block2.conv2", "s_block2.conv2", "s_block1.conv1", "s_block1.conv2")


def resolve_plan3d(precise, dtype: str):
    """None / False -> None (the 16-bit engine); "auto" / "mixed" -> MIXED3D_XW (bf16: every conv); True / "full" -> every conv
    "xw"; a {stage: mode} dict -> that plan.  The transposed convs run "1" (hi plane only) in every plan."""
    if precise is None or precise is False or precise == 0 or precise == "0" or precise == "":
        return None
    convs = [s_ for s_ in STAGES3D if not s_.endswith("upconv1")]
    if precise is True or precise == 1 or precise == "1" or precise == "full":
        plan = {s_: "xw" for s_ in convs}
    elif precise in ("mixed", "auto"):
        from ..unet.unet_engine import MIXED_SEG      # "xw" (default) or "q": the FP8 correction segment (GSSEG_MIXED_SEG)
        plan = {s_: ("xw" if dtype != "f16" else (MIXED_SEG if s_ in MIXED3D_XW else "1")) for s_ in convs}
        if dtype == "f16" and MIXED_SEG == "xw" and os.environ.get("GSSEG_MIXED3D_UPW", "res") == "res":
            # the decoder-entry convs run their w_lo segment on the residual channels only ("xw-": the up half of their input has no
            # lo plane either, so its weight rounding is of the size of the activation rounding the plan already accepts there):
            # K = cin + 2 cr instead of 2 cin + cr -- 448 -> 320 / 896 -> 640 -- 21.3 -> 20.2 ms per 128^3 volume, max |dlogit| over the
            # fixtures 5.4e-4 -> 4.7e-4 (16^3: 4.3e-4 -> 4.1e-4, 5.4e-4 -> 4.7e-4; 128^3: 3.7e-4 -> 4.3e-4).  GSSEG_MIXED3D_UPW=all: all channels.
            for s_ in ("s_block2.conv1", "s_block1.conv1"):
                if plan[s_] == "xw":
                    plan[s_] = "xw-"
    elif isinstance(precise, dict):
        bad = [k for k, v in precise.items() if k not in STAGES3D or v not in ("1", "x", "w", "xw", "xw-", "q")]
        if bad:
            raise ValueError(f"precise plan: unknown stages / modes {bad}")
        plan = {s_: precise.get(s_, "1") for s_ in convs}
    else:
        raise ValueError("precise must be False, True, 'mixed', 'auto' or a {stage: mode} dict")
    for s_ in STAGES3D:
        if s_.endswith("upconv1"):
            plan[s_] = "1"
    return plan


def segs3d(mode: str, cin: int, lo0: int = 0, lo_len=None):
    """(segment list for ops.pack_weight_segs, K extent, input wrap) of a Conv3d stage of the pair forward whose input holds the hi
    plane (cin channels) followed by a lo plane of which the first lo_len channels are valid and belong to the layer's input
    channels [lo0, lo0 + lo_len) (a concat buffer stores [up_h res_h | res_l ...]: lo0 = the up channels, lo_len = the residual
    ones; a plain pair: lo0 = 0, lo_len = cin).  Kinds: 0 = hi(w), 1 = lo(w), 2 = zeros (pads K / the wrap to multiples of 64: the
    padded K chunk multiplies channels of the lo plane by zero, which therefore must hold finite values)."""
    ll = cin if lo_len is None else lo_len
    segs = [(0, 0, cin)]
    span = cin                                   # input channels the K extent walks before it wraps
    if "x" in mode:
        segs.append((0, lo0, ll))
        span += ll
    if span % 64:
        pad = 64 - span % 64
        if pad > ll:
            raise NotImplementedError(f"pair forward: cannot pad a {cin}-channel input to a multiple of 64 channels")
        segs.append((2, 0, pad))
        span += pad
    if mode == "xw-":                            # w_lo only on the channels whose x_lo is valid; the kernel wraps to channel lo0 (wrap_to)
        if lo0 % 64 or ll % 64:
            raise NotImplementedError("pair forward: the partial w_lo segment needs 64-channel-aligned residual channels")
        segs.append((1, lo0, ll))
    elif "w" in mode:
        segs.append((1, 0, cin))
        if cin % 64:
            segs.append((2, 0, 64 - cin % 64))
    K = sum(s_[2] for s_ in segs)
    if len(segs) > 4 or K % 64 or span % 64 or not (span <= K <= 2 * span):
        raise NotImplementedError(f"pair forward: no segment layout for cin={cin}, lo_len={ll}, mode={mode}")
    return segs, K, span


def geom_conv3d(NB, D, H, W, Cin, Cout, dgrad=False, **kw):
    if not dgrad:
        taps = [(ky - 1, kx - 1) for (_, ky, kx) in K3]
        dz = [kd - 1 for (kd, _, _) in K3]
        return ops.make_geom(NB, H, W, Cin, H, W, Cout, H, W, taps, tap_dz=dz, Dg=D, Din=D, Dout=D, **kw)
    taps = [(1 - ky, 1 - kx) for (_, ky, kx) in K3]
    dz = [1 - kd for (kd, _, _) in K3]
    return ops.make_geom(NB, H, W, Cout, H, W, Cin, H, W, taps, tap_dz=dz, Dg=D, Din=D, Dout=D, **kw)


class _Stage:
    __slots__ = ("conv", "bn", "inp", "in_stride", "in_coff", "y", "coef", "stats", "geom", "wd", "cin", "cout",
                 "D", "H", "W", "first", "x3", "halo", "wide")


class UNet3DEngine:
    def __init__(self, net, dtype="f16", precise=None):
        if dtype not in _TORCH_DT:
            raise ValueError("dtype must be 'f16' or 'bf16'")
        self.net, self.dtype, self.tdt = net, dtype, _TORCH_DT[dtype]
        # numerics mode (as UNetEngine): None = single 16-bit storage (logits ~2.4e-3 from the fp32 reference); a plan = the pair
        # forward (forward_pair): every tensor a hi/lo pair, correction MFMA segments where the plan says so.  "auto" falls back
        # to the 16-bit engine for configurations the pair forward does not cover.
        self.plan = resolve_plan3d(precise, dtype)
        self.auto = precise == "auto"
        # data-parallel hooks (parallel.GradReducer.attach): gradients announced as they become final, the compute stream
        # waits for the collectives at the end of backward, autograd receives the reduced gradients
        self.grad_ready_hook = None
        self.after_backward = None
        self.grad_fetch = None

    # ------------------------------------------------------------------------------------------------
    def forward(self, x, training, need_grad):
        net, tdt = self.net, self.tdt
        if not x.is_cuda:
            raise RuntimeError("UNet3D (semantic_segmentation_amd) runs on the MI355X only (no CPU / ATen fallback)")
        if x.dim() != 5 or x.shape[1] != net.in_channels:
            raise ValueError(f"expected input [N,{net.in_channels},D,H,W], got {tuple(x.shape)}")
        NB, _, D0, H0, W0 = x.shape
        if D0 % 8 or H0 % 8 or W0 % 8:
            raise ValueError("volume dims must be multiples of 8 (three 2x2x2 poolings, no padding in the reference)")
        if self.plan is not None:
            if self.auto:
                try:
                    self._pair_layout(W0)
                    covered = net.in_channels == 1 and net.num_classes <= 4 and ops.USE_HALO_CONV
                except NotImplementedError:
                    covered = False
                if covered:
                    return self.forward_pair(x, training, need_grad)
            else:
                return self.forward_pair(x, training, need_grad)
        dev = x.device
        x = x.contiguous().float()

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        stages: List[_Stage] = []

        def bn_coeffs(bn, bias, part, ntiles, C, count):
            coef = empty(4, C, dtype=torch.float32)
            batch = training or bn.running_mean is None
            if batch:
                if training and bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
                upd = training and bn.running_mean is not None
                ops.bn_finalize(part, ntiles, C, count, bn.weight.detach(), bn.bias.detach(),
                                bn.running_mean if upd else None, bn.running_var if upd else None, mom, bn.eps,
                                coef[0], coef[1], coef[2], coef[3])
                if upd and bias is not None:       # statistics were taken before the bias: mean(y + b) = mean(y) + b
                    bn.running_mean.add_(bias.detach(), alpha=mom)
            else:
                rm = bn.running_mean if bias is None else (bn.running_mean - bias.detach())
                ops.bn_eval_coeffs(C, bn.weight.detach(), bn.bias.detach(), rm.contiguous(), bn.running_var, bn.eps,
                                   coef[0], coef[1], coef[2], coef[3])
            return coef, batch

        def conv_bn_relu(conv, bn, inp, in_stride, in_coff, cin, D, H, W, z, z_stride, z_coff, first=False):
            cout = conv.out_channels
            st = _Stage()
            st.conv, st.bn, st.cin, st.cout, st.D, st.H, st.W, st.first = conv, bn, cin, cout, D, H, W, first
            st.inp, st.in_stride, st.in_coff, st.wide = inp, in_stride, in_coff, 0
            y = empty(NB * D, H, W, cout)
            batch = training or bn.running_mean is None
            if first:
                # depth-unfolded view of the volume: X3[nb*D + d][kd] = x[nb][0][d + kd - 1] (zero outside)
                xp = F.pad(inp[:, 0], (0, 0, 0, 0, 1, 1))
                x3 = xp.unfold(1, 3, 1).permute(0, 1, 4, 2, 3).reshape(NB * D, 3, H, W).contiguous()
                st.x3 = x3
                nt = ops.conv_smallcin_mtiles(NB * D, H, W)
                part = empty(ops.bn_partials_numel(nt, cout), dtype=torch.float32) if batch else None
                ops.conv_smallcin_fwd(x3, conv.weight.detach().reshape(cout, 3, 3, 3).contiguous(), None, y, part, 3, 1, 1)
                st.geom = st.wd = None
                st.halo = False
            else:
                w4 = conv.weight.detach().reshape(cout, conv.weight.shape[1], 27, 1)
                if w4.shape[1] != cin:             # a multi-channel input volume: input channels zero-padded to cin = ceil8(C)
                    st.wide = w4.shape[1]
                    wp = torch.zeros((cout, cin, 27, 1), dtype=torch.float32, device=dev)
                    wp[:, :st.wide] = w4
                    w4 = wp
                wf = empty(27, cout, cin)
                wd = empty(27, cin, cout) if need_grad else None
                ops.pack_weight(w4, wf, wd, False)
                st.halo = ops.conv3d3_eligible(cin, cout)
                if st.halo:                      # halo-reuse kernel: (depth tap, channel chunk) K stages
                    g = None
                    nt = ops.conv3d3_stat_rows(NB, D, H, W, cin, cout)
                    part = empty(ops.bn_partials_numel(nt, cout), dtype=torch.float32) if batch else None
                    ops.conv3d3(inp, wf, y, NB, D, H, W, cin, cout, bn_partials=part, in_stride=in_stride, in_coff=in_coff)
                else:
                    g = geom_conv3d(NB, D, H, W, cin, cout, in_stride=in_stride, in_coff=in_coff)
                    nt = ops.conv_igemm_mtiles(g)
                    part = empty(ops.bn_partials_numel(nt, cout), dtype=torch.float32) if batch else None
                    ops.conv_igemm(g, inp, wf, y, None, part)
                st.geom, st.wd = g, wd
            st.coef, st.stats = bn_coeffs(bn, conv.bias, part, nt, cout, NB * D * H * W)
            ops.bn_act_apply(y, st.coef[0], st.coef[1], ACT_RELU, z, z_stride, z_coff)
            st.y = y
            if need_grad:
                stages.append(st)
            return st

        a_blocks = [net.a_block1, net.a_block2, net.a_block3]
        s_blocks = {3: net.s_block3, 2: net.s_block2, 1: net.s_block1}
        dims = [(D0 >> k, H0 >> k, W0 >> k) for k in range(4)]
        cats, cup = {}, {}
        for k in (1, 2, 3):
            sb = s_blocks[k]
            cup[k] = sb.upconv1.out_channels
            cres = a_blocks[k - 1].conv2.out_channels
            D, H, W = dims[k - 1]
            cats[k] = empty(NB * D, H, W, cup[k] + cres)

        # ---- analysis path ----
        enc = []
        inp, in_stride, cin = x, None, net.in_channels
        wide_in = cin != 1
        if wide_in:
            # more than one input channel (the reference takes any, unet3d.py:103-106): the volume enters the MFMA path as a
            # 16-bit NDHWC tensor zero-padded to a multiple of 8 channels (the direct first-layer kernel folds the three depth
            # taps of a ONE-channel volume into its 1..4 input channels)
            if not ops.USE_HALO_CONV:
                raise NotImplementedError("UNet3D with in_channels != 1 needs the halo-reuse conv kernel")
            cpad = (cin + 7) // 8 * 8
            xin = (torch.zeros if cpad != cin else torch.empty)((NB * D0, H0, W0, cpad), dtype=tdt, device=dev)
            ops.nchw_to_nhwc(x.permute(0, 2, 1, 3, 4).reshape(NB * D0, cin, H0, W0).contiguous(), xin, dst_stride=cpad)
            inp, cin = xin, cpad
        for k, blk in enumerate(a_blocks, 1):
            D, H, W = dims[k - 1]
            cmid, cout = blk.conv1.out_channels, blk.conv2.out_channels
            z1 = empty(NB * D, H, W, cmid)
            s1 = conv_bn_relu(blk.conv1, blk.bn1, inp, cin if (k > 1 or wide_in) else None, 0, cin, D, H, W, z1, cmid, 0,
                              first=(k == 1 and not wide_in))
            ctot = cats[k].shape[3]
            s2 = conv_bn_relu(blk.conv2, blk.bn2, z1, cmid, 0, cmid, D, H, W, cats[k], ctot, cup[k])
            pooled = empty(NB * (D // 2), H // 2, W // 2, cout)
            ops.maxpool3d_fwd(cats[k], pooled, NB, D, H, W, cout, ctot, cup[k])
            enc.append((s1, s2, pooled))
            inp, cin = pooled, cout
        D, H, W = dims[3]
        bb = net.bottleNeck
        zb1 = empty(NB * D, H, W, bb.conv1.out_channels)
        sb1 = conv_bn_relu(bb.conv1, bb.bn1, inp, cin, 0, cin, D, H, W, zb1, bb.conv1.out_channels, 0)
        zb = empty(NB * D, H, W, bb.conv2.out_channels)
        sb2 = conv_bn_relu(bb.conv2, bb.bn2, zb1, bb.conv1.out_channels, 0, bb.conv1.out_channels, D, H, W, zb,
                           bb.conv2.out_channels, 0)

        # ---- synthesis path ----
        ups = {}
        cur, ccur = zb, bb.conv2.out_channels
        for k in (3, 2, 1):
            sb = s_blocks[k]
            d, h, w = dims[k]
            D, H, W = dims[k - 1]
            cu, ctot = cup[k], cats[k].shape[3]
            wt = sb.upconv1.weight.detach().reshape(ccur, cu, 8, 1)
            wf = empty(8, cu, ccur)
            wd = empty(8, ccur, cu) if need_grad else None
            ops.pack_weight(wt, wf, wd, True)
            bias = sb.upconv1.bias.detach()
            # all eight sub-voxel classes in one pointwise GEMM (the input volume is read once)
            ops.upconv2x2_fwd(cur, wf, bias, cats[k], NB, d, h, w, ccur, cu, D, H, W, out_stride=ctot, out_coff=0)
            if need_grad:
                taps = [((c >> 1) & 1, c & 1) for c in range(8)]
                gb = ops.make_geom(NB, H, W, cu, h, w, ccur, h, w, taps, isy=2, isx=2, in_stride=ctot, in_coff=0,
                                   tap_dz=[c >> 2 for c in range(8)], Dg=d, Din=D, Dout=d, isz=2)
                ups[k] = dict(zin=cur, wd=wd, geom=gb, cin=ccur, cup=cu, ctot=ctot, dims=(d, h, w, D, H, W))
            cmid = sb.conv1.out_channels
            z1 = empty(NB * D, H, W, cmid)
            conv_bn_relu(sb.conv1, sb.bn, cats[k], ctot, 0, ctot, D, H, W, z1, cmid, 0)
            z2 = empty(NB * D, H, W, cmid)
            conv_bn_relu(sb.conv2, sb.bn, z1, cmid, 0, cmid, D, H, W, z2, cmid, 0)
            cur, ccur = z2, cmid
        head = net.s_block1.conv3
        ncls = head.out_channels
        if ncls > 4:
            raise NotImplementedError("num_classes above 4 is not supported by the direct head kernel")
        l2d = empty(NB * D0, ncls, H0, W0, dtype=torch.float32)
        ops.conv_smallcout_fwd(cur, head.weight.detach().reshape(ncls, ccur, 1, 1).contiguous(), head.bias.detach(), l2d)
        logits = l2d.view(NB, D0, ncls, H0, W0).permute(0, 2, 1, 3, 4).contiguous()
        ctx = None
        if need_grad:
            ctx = dict(stages=stages, enc=enc, bott=(sb1, sb2), ups=ups, cats=cats, cup=cup, dims=dims, NB=NB,
                       z_last=cur, c_last=ccur, ncls=ncls)
        return logits, ctx

    # ------------------------------------------------------------------------------------------------
    def _pair_layout(self, W0: int = 128):
        """segment layouts of every conv stage under self.plan (raises NotImplementedError for channel counts the pair forward
        cannot lay out): ({stage: (segs, K, wrap)}, the effective plan); cached per (plan, volume width).  A "q" stage (FP8
        correction segment) needs the LDS-DMA kernel -- level width >= 24, channels % 64 == 0, fp16 -- and runs as "xw" elsewhere;
        its entry is ("q", 2 * cin, 2 * cin)."""
        cache = self.__dict__.setdefault("_lay", {})
        if W0 in cache:
            return cache[W0]
        net = self.net
        plan = dict(self.plan)
        level = {"a_block1": 0, "a_block2": 1, "a_block3": 2, "bottleNeck": 3, "s_block3": 2, "s_block2": 1, "s_block1": 0}
        for st_, m_ in self.plan.items():
            if m_ == "q":
                conv = net.get_submodule(st_)
                ok = (not st_.endswith("upconv1") and self.dtype == "f16"
                      and ops.conv3x3_q8_ok(W0 >> level[st_.split(".")[0]], conv.in_channels, conv.out_channels))
                if not ok:
                    plan[st_] = "xw"
        lay = {}
        for k, blk in (("a_block1", net.a_block1), ("a_block2", net.a_block2), ("a_block3", net.a_block3), ("bottleNeck", net.bottleNeck)):
            def seg_or_q(name, cin, **kw):
                return ("q", 2 * cin, 2 * cin) if plan[name] == "q" else segs3d(plan[name], cin, **kw)
            if k != "a_block1":
                lay[k + ".conv1"] = seg_or_q(k + ".conv1", blk.conv1.in_channels)
            elif net.in_channels != 1:
                raise NotImplementedError("pair forward: one input channel (the direct first-layer kernel)")
            lay[k + ".conv2"] = seg_or_q(k + ".conv2", blk.conv2.in_channels)
        for k, sb, ab in (("s_block3", net.s_block3, net.a_block3), ("s_block2", net.s_block2, net.a_block2),
                          ("s_block1", net.s_block1, net.a_block1)):
            cu, cr = sb.upconv1.out_channels, ab.conv2.out_channels
            lay[k + ".conv1"] = ("q", 2 * (cu + cr), 2 * (cu + cr)) if plan[k + ".conv1"] == "q" else segs3d(plan[k + ".conv1"], cu + cr, lo0=cu, lo_len=cr)
            lay[k + ".conv2"] = ("q", 2 * sb.conv2.in_channels, 2 * sb.conv2.in_channels) if plan[k + ".conv2"] == "q" else \
                segs3d(plan[k + ".conv2"], sb.conv2.in_channels)
        if net.s_block1.conv2.out_channels != 64:
            raise NotImplementedError("pair forward: the pointwise head kernel reads 64 channels")
        cache[W0] = (lay, plan)
        return cache[W0]

    def forward_pair(self, x, training, need_grad):
        """The pair forward of UNet3D (BASELINE config 5 at the north star's 1e-3 on logits; GenSeg-3D/UNet3D/unet3d.py:89-126 is
        plain fp32): every activation and conv output travels as a PAIR of 16-bit values v = hi + lo; BatchNorm / ReLU / max-pool /
        the head read and write pairs; a conv stage runs the MFMA segments of self.plan (unet_engine.forward_precise explains the
        forms).  The transposed convs run on the LDS-DMA GEMM and write the hi plane only; the first conv reads the fp32 volume.
        HBM layout: pair buffers [NB*D, H, W, 2*C] = [hi (C) | lo (C)]; concat buffers [.., 2*(cup + cres)] =
        [up_h res_h | res_l -] -- the residual's lo plane FIRST, so that the consumer's x_lo segment (residual channels only) is
        contiguous behind the hi plane.  The backward pass is the default one: it reads the hi planes through strides."""
        net, tdt = self.net, self.tdt
        if not ops.USE_HALO_CONV:
            raise RuntimeError("the pair forward needs the halo-reuse conv kernels (GSSEG_CONV3X3=halo)")
        NB, _, D0, H0, W0 = x.shape
        lay, plan = self._pair_layout(W0)              # plan: "q" stages this volume size has no kernel for run as "xw"

        def lo_fmt(st_name):
            """form in which conv stage st_name reads the lo plane of its input: 1 = 16-bit lo plane, 2 = q plane (FP8 chunks)"""
            return 2 if plan[st_name] == "q" else 1
        dev = x.device
        x = x.contiguous().float()
        names = {id(m): n for n, m in net.named_modules()}

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        # every stale segment pack in ONE launch
        items, qitems, packs = [], [], {}
        for st_name, (segs, K, _) in lay.items():
            conv = net.get_submodule(st_name)
            cout, cin = conv.out_channels, conv.in_channels
            pack = empty(27, cout, K)
            if segs == "q":
                wexp = torch.empty(cout, dtype=torch.int32, device=dev)
                qitems.append((conv.weight.detach().reshape(cout, cin, 27, 1), pack, wexp))
                packs[st_name] = (pack, wexp)
            else:
                items.append((conv.weight.detach().reshape(cout, cin, 27, 1), pack, False, segs))
                packs[st_name] = pack
        if items:
            ops.pack_weight_segs(items)
        if qitems:
            ops.pack_weight_q8(qitems)
        dpacks = {}
        if need_grad:                              # the backward's data-gradient packs (16-bit, as the default engine): one launch
            ditems = []
            for st_name in lay:
                conv = net.get_submodule(st_name)
                cout, cin = conv.out_channels, conv.in_channels
                dpacks[st_name] = empty(27, cin, cout)
                ditems.append((conv.weight.detach().reshape(cout, cin, 27, 1), None, dpacks[st_name], False))
            ops.pack_weight_multi(ditems)

        stages: List[_Stage] = []

        def bn_coeffs(bn, bias, part, ntiles, C, count):
            coef = empty(4, C, dtype=torch.float32)
            batch = training or bn.running_mean is None
            if batch:
                if training and bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
                upd = training and bn.running_mean is not None
                ops.bn_finalize(part, ntiles, C, count, bn.weight.detach(), bn.bias.detach(),
                                bn.running_mean if upd else None, bn.running_var if upd else None, mom, bn.eps,
                                coef[0], coef[1], coef[2], coef[3])
                if upd and bias is not None:       # statistics were taken before the bias: mean(y + b) = mean(y) + b
                    bn.running_mean.add_(bias.detach(), alpha=mom)
            else:
                rm = bn.running_mean if bias is None else (bn.running_mean - bias.detach())
                ops.bn_eval_coeffs(C, bn.weight.detach(), bn.bias.detach(), rm.contiguous(), bn.running_var, bn.eps,
                                   coef[0], coef[1], coef[2], coef[3])
            return coef, batch

        def stage(conv, bn, inp, in_stride, cin, D, H, W, z_hi, z_lo, z_stride, z_coff, first=False, z_q8=False, zp=None):
            """conv (+ bias folded into BatchNorm) -> BN -> ReLU on pairs; the z pair goes to z_hi / z_lo (views; both take
            z_coff) with pixel stride z_stride; z_lo None: not stored; z_q8: z_lo is the buffer's q plane (its byte 0); zp =
            (zp_hi, zp_lo, zp_stride): MaxPool3d(2) of the z pair in the same pass (16-bit lo planes only)."""
            cout = conv.out_channels
            st = _Stage()
            st.conv, st.bn, st.cin, st.cout, st.D, st.H, st.W, st.first = conv, bn, cin, cout, D, H, W, first
            st.inp, st.in_stride, st.in_coff, st.wide, st.geom, st.x3 = inp, in_stride, 0, 0, None, None
            y_hi, y_lo = empty(NB * D, H, W, cout), empty(NB * D, H, W, cout)
            batch = training or bn.running_mean is None
            if first:
                xp = F.pad(inp[:, 0], (0, 0, 0, 0, 1, 1))
                x3 = xp.unfold(1, 3, 1).permute(0, 1, 4, 2, 3).reshape(NB * D, 3, H, W).contiguous()
                st.x3 = x3
                nt = ops.conv_smallcin_mtiles(NB * D, H, W)
                part = empty(ops.bn_partials_numel(nt, cout), dtype=torch.float32) if batch else None
                ops.conv_smallcin_fwd_split(x3, conv.weight.detach().reshape(cout, 3, 3, 3).contiguous(), y_hi, y_lo, part, 3, 1)
                st.wd, st.halo = None, False
            else:
                segs_, K, wrap = lay[names[id(conv)]]
                nt = ops.conv3x3_stat_rows(NB * D, H, W, K, cout, pair="q" if segs_ == "q" else True)
                part = empty(ops.bn_partials_numel(nt, cout), dtype=torch.float32) if batch else None
                if segs_ == "q":
                    qpack, wexp = packs[names[id(conv)]]
                    ops.conv3d3_q8(inp, qpack, wexp, y_hi, y_lo, NB, D, H, W, cin, cout, in_stride, 0, part)
                else:
                    wto = (conv.in_channels - (wrap - conv.in_channels)) if plan[names[id(conv)]] == "xw-" else 0     # lo0 = cin - lo_len
                    ops.conv3d3_segs(inp, packs[names[id(conv)]], y_hi, y_lo, NB, D, H, W, K, wrap, cin, cout, in_stride, 0, part, wrap_to=wto)
                st.halo = True
                st.wd = dpacks.get(names[id(conv)])
            st.coef, st.stats = bn_coeffs(bn, conv.bias, part, nt, cout, NB * D * H * W)
            if z_q8:
                ops.bn_act_apply_split_q8(y_hi, y_lo, st.coef[0], st.coef[1], ACT_RELU, z_hi, z_lo, True, z_stride, z_coff)
            elif zp is not None:
                ops.bn_act_apply_split_pool3d(y_hi, y_lo, st.coef[0], st.coef[1], ACT_RELU, z_hi, z_lo, z_stride, z_coff, zp[0], zp[1], zp[2],
                                              NB, D, H, W)
            else:
                ops.bn_act_apply_split(y_hi, y_lo, st.coef[0], st.coef[1], ACT_RELU, z_hi, z_lo, z_stride, z_coff)
            st.y = y_hi
            if need_grad:
                stages.append(st)
            return st

        a_blocks = [net.a_block1, net.a_block2, net.a_block3]
        s_blocks = {3: net.s_block3, 2: net.s_block2, 1: net.s_block1}
        dims = [(D0 >> k, H0 >> k, W0 >> k) for k in range(4)]
        cats, cup = {}, {}
        for k in (1, 2, 3):
            cup[k] = s_blocks[k].upconv1.out_channels
            cres = a_blocks[k - 1].conv2.out_channels
            D, H, W = dims[k - 1]
            # the unused tail of the lo plane (the up channels) is never read: a consumer's x_lo / zero segments stop at cres
            cats[k] = empty(NB * D, H, W, 2 * (cup[k] + cres))

        # ---- analysis path ----
        enc = []
        inp, in_stride, cin = x, None, 1
        for k, blk in enumerate(a_blocks, 1):
            D, H, W = dims[k - 1]
            cmid, cout = blk.conv1.out_channels, blk.conv2.out_channels
            z1 = empty(NB * D, H, W, 2 * cmid)
            s1 = stage(blk.conv1, blk.bn1, inp, in_stride, cin, D, H, W, z1, z1[..., cmid:], 2 * cmid, 0, first=(k == 1),
                       z_q8=lo_fmt(f"a_block{k}.conv2") == 2)
            ctot = cup[k] + cout
            cat = cats[k]
            cat_q8 = lo_fmt(f"s_block{k}.conv1") == 2
            if cat_q8:
                # residual: hi plane at channels [cup, ctot); the buffer's second half is the q plane of ALL ctot channels
                s2 = stage(blk.conv2, blk.bn2, z1, 2 * cmid, cmid, D, H, W, cat, cat[..., ctot:], 2 * ctot, cup[k], z_q8=True)
            pooled = empty(NB * (D // 2), H // 2, W // 2, 2 * cout)
            nxt_conv = f"a_block{k + 1}.conv1" if k < 3 else "bottleNeck.conv1"
            fused_pool = FUSED_POOL3D and not cat_q8 and lo_fmt(nxt_conv) != 2 and D % 2 == 0 and H % 2 == 0 and W % 2 == 0
            if not cat_q8:
                # residual: hi plane at channels [cup, ctot), 16-bit lo plane at [ctot, ctot + cout); the pooled pair in the same pass
                s2 = stage(blk.conv2, blk.bn2, z1, 2 * cmid, cmid, D, H, W, cat, cat[..., ctot - cup[k]:], 2 * ctot, cup[k],
                           zp=(pooled, pooled[..., cout:], 2 * cout) if fused_pool else None)
            if not fused_pool:
                ops.maxpool3d_fwd_pair_q8(cat[..., cup[k]:], cat[..., ctot:], cat_q8, cup[k], 2 * ctot, pooled, pooled[..., cout:],
                                          lo_fmt(nxt_conv) == 2, 2 * cout, NB, D, H, W, cout)
            enc.append((s1, s2, pooled))
            inp, in_stride, cin = pooled, 2 * cout, cout
        D, H, W = dims[3]
        bb = net.bottleNeck
        c1, c2 = bb.conv1.out_channels, bb.conv2.out_channels
        zb1 = empty(NB * D, H, W, 2 * c1)
        sb1 = stage(bb.conv1, bb.bn1, inp, in_stride, cin, D, H, W, zb1, zb1[..., c1:], 2 * c1, 0, z_q8=lo_fmt("bottleNeck.conv2") == 2)
        zb = empty(NB * D, H, W, 2 * c2)
        sb2 = stage(bb.conv2, bb.bn2, zb1, 2 * c1, c1, D, H, W, zb, None, 2 * c2, 0)       # one reader: the hi-only transposed conv

        # ---- synthesis path ----
        ups = {}
        cur, ccur = zb, c2
        z_last = None
        for k in (3, 2, 1):
            sb = s_blocks[k]
            d, h, w = dims[k]
            D, H, W = dims[k - 1]
            cu = cup[k]
            ctot = cats[k].shape[3] // 2
            wt = sb.upconv1.weight.detach().reshape(ccur, cu, 8, 1)
            wf = empty(8, cu, ccur)
            wd = empty(8, ccur, cu) if need_grad else None
            ops.pack_weight(wt, wf, wd, True)
            # x_hi . w_hi on the LDS-DMA pointwise GEMM, hi plane only (all eight sub-voxel classes in one launch)
            ops.upconv2x2_fwd(cur, wf, sb.upconv1.bias.detach(), cats[k], NB, d, h, w, ccur, cu, D, H, W, in_stride=2 * ccur,
                              out_stride=2 * ctot, out_coff=0)
            if lo_fmt(f"s_block{k}.conv1") == 2:       # a "q" consumer: the q chunks of the up channels (hi8 from the hi plane, lo8 = 0)
                ops.q8_from_hi(cats[k], cats[k][..., ctot:], NB * D * H * W, cu, 2 * ctot, 0)
            if need_grad:
                taps = [((c >> 1) & 1, c & 1) for c in range(8)]
                tdz = [c >> 2 for c in range(8)]
                # the transposed conv seen from its output side: dU = the up channels of the DENSE gradient of the concat input
                gb = ops.make_geom(NB, H, W, cu, h, w, ccur, h, w, taps, isy=2, isx=2, in_stride=ctot, in_coff=0,
                                   tap_dz=tdz, Dg=d, Din=D, Dout=d, isz=2)
                # ... and for the weight gradient, whose second operand is the hi plane of the layer's input pair
                gw = ops.make_geom(NB, H, W, cu, h, w, ccur, h, w, taps, isy=2, isx=2, in_stride=ctot, in_coff=0,
                                   out_stride=2 * ccur, tap_dz=tdz, Dg=d, Din=D, Dout=d, isz=2)
                ups[k] = dict(zin=cur, wd=wd, geom=gb, geom_wg=gw, cin=ccur, cup=cu, ctot=ctot, dims=(d, h, w, D, H, W))
            cmid = sb.conv1.out_channels
            z1 = empty(NB * D, H, W, 2 * cmid)
            stage(sb.conv1, sb.bn, cats[k], 2 * ctot, ctot, D, H, W, z1, z1[..., cmid:], 2 * cmid, 0, z_q8=lo_fmt(f"s_block{k}.conv2") == 2)
            if k > 1:
                z2 = empty(NB * D, H, W, 2 * cmid)
                stage(sb.conv2, sb.bn, z1, 2 * cmid, cmid, D, H, W, z2, None, 2 * cmid, 0)  # one reader: the next transposed conv
                cur, ccur = z2, cmid
            else:
                zl = empty(2, NB * D, H, W, cmid)                   # two dense planes: the head and its backward read dense tensors
                stage(sb.conv2, sb.bn, z1, 2 * cmid, cmid, D, H, W, zl[0], zl[1], cmid, 0)
                z_last, ccur = zl, cmid
        head = net.s_block1.conv3
        ncls = head.out_channels
        l2d = empty(NB * D0, ncls, H0, W0, dtype=torch.float32)
        ops.head1x1_fwd_split(z_last[0], z_last[1], head.weight.detach().reshape(ncls, ccur).contiguous(), head.bias.detach(), l2d)
        logits = l2d.view(NB, D0, ncls, H0, W0).permute(0, 2, 1, 3, 4).contiguous()
        ctx = None
        if need_grad:
            ctx = dict(stages=stages, enc=enc, bott=(sb1, sb2), ups=ups, cats=cats, cup=cup, dims=dims, NB=NB,
                       z_last=z_last[0], c_last=ccur, ncls=ncls, pair=True)
        return logits, ctx

    # ------------------------------------------------------------------------------------------------
    def backward(self, ctx, dlogits):
        net, tdt = self.net, self.tdt
        NB, dims, cats, cup, ups = ctx["NB"], ctx["dims"], ctx["cats"], ctx["cup"], ctx["ups"]
        dev = dlogits.device
        D0, H0, W0 = dims[0]
        S = float(2 ** round(math.log2(NB * D0 * H0 * W0)))
        inv_s = 1.0 / S
        names = {id(p): n for n, p in net.named_parameters()}
        # a gradient is announced to the reducer (bucketed all-reduce overlapped with the rest of backward) once ALL its
        # contributions are in: the decoder blocks' shared BatchNorm3d is used by two stages
        from ..parallel import GradEmitter
        expected: Dict[str, int] = {}
        for st_ in ctx["stages"]:
            for p_ in (st_.bn.weight, st_.bn.bias):
                expected[names[id(p_)]] = expected.get(names[id(p_)], 0) + 1
        emitter = GradEmitter(self.grad_ready_hook, expected)
        grads = emitter.grads

        def empty(*shape, dtype=tdt):
            return torch.empty(shape, dtype=dtype, device=dev)

        def emit(p, g):
            emitter.emit(names[id(p)], g)

        col_ws = empty(1024 * 1024, dtype=torch.float32)

        def stage_bwd(st: _Stage, dz, dz_stride, dz_coff, need_dinp):
            D, H, W, cin, cout = st.D, st.H, st.W, st.cin, st.cout
            n2 = NB * D
            nt = ops.bn_bwd_tiles_used(n2, H, W, False)
            part = empty(ops.bn_partials_numel(ops.bn_bwd_tiles(n2, H, W), cout), dtype=torch.float32)
            c = st.coef
            ops.bn_act_bwd_reduce(st.y, dz, dz_stride, dz_coff, None, c[0], c[1], c[2], c[3], ACT_RELU, part)
            dgamma, dbeta = empty(cout, dtype=torch.float32), empty(cout, dtype=torch.float32)
            c12 = empty(2, cout, dtype=torch.float32)
            ops.bn_bwd_coeffs(part, nt, cout, n2 * H * W, inv_s, dgamma, dbeta, c12[0], c12[1])
            if not st.stats:
                c12.zero_()
            dy = empty(n2, H, W, cout)
            ops.bn_act_bwd_apply(st.y, dz, dz_stride, dz_coff, None, c[0], c[1], c[2], c[3], c12[0], c12[1], ACT_RELU,
                                 True, dy)
            emit(st.bn.weight, dgamma)
            emit(st.bn.bias, dbeta)
            if st.conv.bias is not None:
                if st.stats:
                    # a bias in front of a batch-statistics BatchNorm has the gradient sum_p dy = sc * (sum gh - N mean(gh)
                    # - mean(gh xh) * sum_p xh) = 0 exactly (sum_p xh = 0): the reference's value is fp32 rounding noise (~1e-8 of
                    # |dy|); a column sum over the 16-bit dy would be 16-bit rounding noise and cost a pass over dy per stage
                    # (17 passes, 0.68 ms of the 18.7 ms 128^3 step)
                    db = torch.zeros(cout, dtype=torch.float32, device=dev)
                else:
                    db = empty(cout, dtype=torch.float32)
                    ops.colsum(dy, cout, 0, n2, H, W, 0, 0, H, W, cout, inv_s, col_ws, db)
                emit(st.conv.bias, db)
            wparam = st.conv.weight
            dinp = None
            if st.first:
                dw = torch.zeros((cout, 3, 3, 3), dtype=torch.float32, device=dev)
                ops.conv_smallcin_wgrad(st.x3, dy, dw, 3, 1, 1, inv_s)
                emit(wparam, dw.view(wparam.shape))
            else:
                dw = torch.empty((cout, cin, 27), dtype=torch.float32, device=dev)
                if st.halo:
                    # split-K parts in slabs + ordered reduction fused with scale / unpack: deterministic, no atomics
                    wsl = empty(ops.conv3d3_wgrad_ws_floats(NB, D, H, W, cin, cout), dtype=torch.float32)
                    ops.conv3d3_wgrad_det(st.inp, dy, wsl, dw, NB, D, H, W, cin, cout, inv_s, in_stride=st.in_stride,
                                          in_coff=st.in_coff)
                else:
                    wsl = empty(ops.conv_wgrad_ws_floats(st.geom), dtype=torch.float32)
                    ops.conv_wgrad_det(st.geom, st.inp, dy, wsl, dw, cout, cin, 27, inv_s)
                if st.wide:                          # the padded input channels carry no parameter
                    dw = dw.view(cout, cin, 27)[:, :st.wide].contiguous()
                emit(wparam, dw.view(wparam.shape))
                if need_dinp:
                    dinp = empty(n2, H, W, cin)
                    if ops.conv3d3_eligible(cout, cin):
                        ops.conv3d3(dy, st.wd, dinp, NB, D, H, W, cout, cin, dgrad=True)
                    else:
                        ops.conv_igemm(geom_conv3d(NB, D, H, W, cin, cout, dgrad=True), dy, st.wd, dinp)
            return dinp

        stages = {id(s): s for s in ctx["stages"]}
        by_conv = {id(s.conv): s for s in ctx["stages"]}
        head = net.s_block1.conv3
        ncls = ctx["ncls"]
        dl = (dlogits.contiguous().float() * S).permute(0, 2, 1, 3, 4).contiguous().view(NB * D0, ncls, H0, W0)
        dwo = torch.zeros((ncls, ctx["c_last"], 1, 1), dtype=torch.float32, device=dev)
        dbo = torch.zeros(ncls, dtype=torch.float32, device=dev)
        dz = empty(NB * D0, H0, W0, ctx["c_last"])
        ops.conv_smallcout_bwd(ctx["z_last"], head.weight.detach().reshape(ncls, ctx["c_last"], 1, 1).contiguous(), dl,
                               dz, dwo, dbo, gscale=inv_s)
        emit(head.weight, dwo.view(head.weight.shape))
        emit(head.bias, dbo)

        s_blocks = {3: net.s_block3, 2: net.s_block2, 1: net.s_block1}
        dres = {}
        for k in (1, 2, 3):
            sb = s_blocks[k]
            st2, st1 = by_conv[id(sb.conv2)], by_conv[id(sb.conv1)]
            dmid = stage_bwd(st2, dz, st2.cout, 0, True)
            dcat = stage_bwd(st1, dmid, st1.cout, 0, True)                 # [NB*D, H, W, cup + cres]
            u = ups[k]
            d, h, w, D, H, W = u["dims"]
            cu, ctot, ccur = u["cup"], u["ctot"], u["cin"]
            dres[k] = dcat
            db = empty(cu, dtype=torch.float32)
            ops.colsum(dcat, ctot, 0, NB * D, H, W, 0, 0, H, W, cu, inv_s, col_ws, db)
            emit(sb.upconv1.bias, db)
            dw = torch.empty((ccur, cu, 8), dtype=torch.float32, device=dev)
            gwg = u.get("geom_wg", u["geom"])              # pair forward: the layer's input is the hi plane of a pair buffer
            wsl = empty(ops.conv_wgrad_ws_floats(gwg), dtype=torch.float32)
            ops.conv_wgrad_det(gwg, dcat, u["zin"], wsl, dw, ccur, cu, 8, inv_s)
            emit(sb.upconv1.weight, dw.view(sb.upconv1.weight.shape))
            dz = empty(NB * d, h, w, ccur)
            ops.conv_igemm(u["geom"], dcat, u["wd"], dz)
        # bottleneck
        sb1, sb2 = ctx["bott"]
        dmid = stage_bwd(sb2, dz, sb2.cout, 0, True)
        dpool = stage_bwd(sb1, dmid, sb1.cout, 0, True)
        # analysis path, deepest first
        for k in (3, 2, 1):
            s1, s2, _ = ctx["enc"][k - 1]
            D, H, W = dims[k - 1]
            zs = cats[k].shape[3]                          # pixel stride of the concat buffer (pair forward: both planes)
            ctot = zs // 2 if ctx.get("pair") else zs
            dz2 = empty(NB * D, H, W, s2.cout)
            ops.maxpool3d_bwd(cats[k], dpool, dres[k], dz2, NB, D, H, W, s2.cout, zs, cup[k], ctot, cup[k])
            dmid = stage_bwd(s2, dz2, s2.cout, 0, True)
            dpool = stage_bwd(s1, dmid, s1.cout, 0, k > 1)
        if self.after_backward is not None:
            self.after_backward()
        return grads


class _UNet3DFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, training, need_grad, x, *plist):
        logits, ectx = engine.forward(x, training, need_grad)
        ctx.engine, ctx.ectx, ctx.plist = engine, ectx, plist
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.ectx is None:
            raise RuntimeError("UNet3D forward ran without gradient tracking")
        grads = ctx.engine.backward(ctx.ectx, dlogits)
        names = [n for n, _ in ctx.engine.net.named_parameters()]
        fetch = ctx.engine.grad_fetch               # data parallel: the reduced gradients (parallel.GradReducer.fetch)
        out = [(fetch(n) if (fetch is not None and n in grads) else grads.get(n)) if p.requires_grad else None
               for n, p in zip(names, ctx.plist)]
        return (None, None, None, None, *out)


def run_unet3d(engine, x):
    plist = tuple(p for p in engine.net.parameters())
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
    return _UNet3DFunction.apply(engine, engine.net.training, need_grad, x, *plist)
