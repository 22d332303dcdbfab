"""Stand-alone execution of the U-Net building blocks (reference: unet/unet_parts.py:8-77 -- `DoubleConv(x)`, `Down(x)`,
`Up(x1, x2)`, `OutConv(x)` are ordinary callable modules there) on the same HIP kernels as the whole-network engine.

Not the hot path: `UNet.forward` runs unet_engine.py, which fuses across block boundaries (pool into the producer's BN pass,
the transposed conv straight into the concat buffer, no layout conversions).  A block called on its own takes and returns
fp32 NCHW tensors like the reference, so it pays a layout conversion on the way in and out; inside it is
conv3x3 (MFMA, halo reuse) -> BN statistics from the conv epilogue -> BN+ReLU pass, with first-order autograd through the
hand-written backward kernels.  Gradients travel in 16 bits times a power-of-two scale derived on the device from the
incoming gradient's amax (no host sync); parameter gradients are multiplied back exactly."""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops
from .._lib import ACT_NONE, ACT_RELU

_TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _tdt():
    import os
    return _TORCH_DT[os.environ.get("GSSEG_DTYPE", "f16")]


def _check(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"{what} (semantic_segmentation_amd) runs on the MI355X only: there is no CPU / ATen fallback")
    if x.dim() != 4:
        raise ValueError(f"{what}: expected an NCHW tensor, got shape {tuple(x.shape)}")


def _to_nhwc(x: torch.Tensor, dst: Optional[torch.Tensor] = None, stride=None, coff=0) -> torch.Tensor:
    N, C, H, W = x.shape
    if dst is None:
        dst = torch.empty((N, H, W, C), dtype=_tdt(), device=x.device)
    ops.nchw_to_nhwc(x.contiguous().float(), dst, stride, coff)
    return dst


def _to_nchw(t16: torch.Tensor, C: int, stride=None, coff=0) -> torch.Tensor:
    N, H, W = t16.shape[0], t16.shape[1], t16.shape[2]
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=t16.device)
    ops.nhwc_to_nchw(t16, out, stride, coff)
    return out


class _Stage:
    """conv3x3(pad 1, no bias) -> BatchNorm -> ReLU of `DoubleConv` (unet_parts.py:16-21), forward state kept for backward."""

    def __init__(self, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d):
        self.conv, self.bn = conv, bn

    def forward(self, inp, image: bool, N, h, w, training: bool, z, z_stride, z_coff, in_stride=None):
        conv, bn = self.conv, self.bn
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        dev, tdt = inp.device, _tdt()
        y = torch.empty((N, h, w, cout), dtype=tdt, device=dev)
        batch_stats = training or bn.running_mean is None
        ntiles = ops.conv_smallcin_mtiles(N, h, w) if image else ops.conv3x3_stat_rows(N, h, w, cin, cout)
        part = torch.empty(ops.bn_partials_numel(ntiles, cout), dtype=torch.float32, device=dev) if batch_stats else None
        self.wd = None
        if image:
            ops.conv_smallcin_fwd(inp, conv.weight.detach().contiguous(), None, y, part, 3, 1, 1)
        else:
            wf = torch.empty((9, cout, cin), dtype=tdt, device=dev)
            self.wd = torch.empty((9, cin, cout), dtype=tdt, device=dev)
            ops.pack_weight(conv.weight.detach().contiguous(), wf, self.wd, False)
            ops.conv3x3(inp, wf, y, N, h, w, cin, cout, ops.TAPS3_FWD, None, part, in_stride=in_stride)
        coef = torch.empty((4, cout), dtype=torch.float32, device=dev)
        if batch_stats:
            mom = bn.momentum
            if training and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            if mom is None:
                mom = 1.0 / float(bn.num_batches_tracked.item()) if bn.num_batches_tracked is not None else 0.0
            upd = training and bn.running_mean is not None
            ops.bn_finalize(part, ntiles, cout, N * h * w, bn.weight.detach(), bn.bias.detach(),
                            bn.running_mean if upd else None, bn.running_var if upd else None, mom, bn.eps,
                            coef[0], coef[1], coef[2], coef[3])
        else:
            ops.bn_eval_coeffs(cout, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                               coef[0], coef[1], coef[2], coef[3])
        ops.bn_act_apply(y, coef[0], coef[1], ACT_RELU, z, z_stride, z_coff)
        self.inp, self.in_stride, self.image, self.y, self.coef, self.stats = inp, in_stride, image, y, coef, batch_stats
        self.N, self.h, self.w, self.cin, self.cout = N, h, w, cin, cout

    def backward(self, dz, sa, ca, need_dinp: bool):
        """dz: 16-bit gradient w.r.t. the stage output (pixel stride sa, channel offset ca).  Returns (dinp, dw, dgamma, dbeta)."""
        N, h, w, cin, cout = self.N, self.h, self.w, self.cin, self.cout
        dev, tdt, coef = dz.device, _tdt(), self.coef
        part = torch.empty(ops.bn_partials_numel(ops.bn_bwd_tiles(N, h, w), cout), dtype=torch.float32, device=dev)
        ops.bn_act_bwd_reduce(self.y, dz, sa, ca, None, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        c12 = torch.empty((2, cout), dtype=torch.float32, device=dev)
        ops.bn_bwd_coeffs(part, ops.bn_bwd_tiles_used(N, h, w, False), cout, N * h * w, 1.0, dgamma, dbeta, c12[0], c12[1])
        if not self.stats:
            c12.zero_()
        dy = torch.empty((N, h, w, cout), dtype=tdt, device=dev)
        ops.bn_act_bwd_apply(self.y, dz, sa, ca, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy)
        wparam = self.conv.weight
        dinp = None
        if self.image:
            dw = torch.zeros_like(wparam, memory_format=torch.contiguous_format)
            ops.conv_smallcin_wgrad(self.inp, dy, dw, 3, 1, 1, 1.0)
            if need_dinp:
                dinp = torch.empty_like(self.inp)
                ops.conv_smallcin_dgrad(dy, wparam.detach().contiguous(), dinp, 3, 1, 1, 1.0)
        else:
            dw = torch.empty_like(wparam, memory_format=torch.contiguous_format)
            ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, h, w, cin, cout), dtype=torch.float32, device=dev)
            ops.conv3x3_wgrad_det(self.inp, dy, ws, dw, N, h, w, cin, cout, 1.0, in_stride=self.in_stride)
            if need_dinp:
                dinp = torch.empty((N, h, w, cin), dtype=tdt, device=dev)
                ops.conv3x3(dy, self.wd, dinp, N, h, w, cout, cin, ops.TAPS3_DGRAD)
        return dinp, dw, dgamma, dbeta


def _grad_scale(g: torch.Tensor) -> torch.Tensor:
    """power of two r (device scalar) with amax(g) * r in [1, 2): what the 16-bit backward kernels are comfortable with"""
    amax = g.detach().abs().amax().float()
    return torch.exp2(torch.clamp(torch.floor(-torch.log2(torch.clamp(amax, min=1e-30))), -60.0, 60.0))


def _double_conv_parts(dc):
    seq = dc.double_conv
    return _Stage(seq[0], seq[1]), _Stage(seq[3], seq[4])


def _dc_params(dc) -> List[torch.Tensor]:
    seq = dc.double_conv
    return [seq[0].weight, seq[1].weight, seq[1].bias, seq[3].weight, seq[4].weight, seq[4].bias]


class _BlockFunction(torch.autograd.Function):
    """kind: 'double' (DoubleConv), 'down' (MaxPool2d(2) + DoubleConv), 'up' (Up: x = x1, skip = x2)."""

    @staticmethod
    def forward(ctx, kind, module, training, x, skip, *params):
        tdt = _tdt()
        dc = module if kind == "double" else (module.maxpool_conv[1] if kind == "down" else module.conv)
        s0, s3 = _double_conv_parts(dc)
        N, Cx, H, W = x.shape
        dev = x.device
        state = {"kind": kind, "s0": s0, "s3": s3, "x_shape": tuple(x.shape)}
        if kind == "double":
            image = Cx <= 4
            if not image and Cx % 8:
                raise NotImplementedError("stand-alone DoubleConv: in_channels must be 1..4 or a multiple of 8")
            inp = x.contiguous().float() if image else _to_nhwc(x)
            h, w, in_stride = H, W, None
        elif kind == "down":
            if Cx % 8:
                raise NotImplementedError("stand-alone Down: in_channels must be a multiple of 8")
            x16 = _to_nhwc(x)
            h, w = H // 2, W // 2
            inp = torch.empty((N, h, w, Cx), dtype=tdt, device=dev)
            ops.maxpool2x2_fwd(x16, inp, N, H, W, Cx)
            image, in_stride = False, None
            state["x16"] = x16
        else:                                                          # Up.forward(x1, x2): unet_parts.py:56-68
            Cs, H2, W2 = skip.shape[1], skip.shape[2], skip.shape[3]
            bilinear = not isinstance(module.up, torch.nn.ConvTranspose2d)
            Cu = Cx if bilinear else module.up.weight.shape[1]
            if Cx % 8 or Cs % 8 or Cu % 8:
                raise NotImplementedError("stand-alone Up: channel counts must be multiples of 8")
            if H2 < 2 * H or W2 < 2 * W:
                raise ValueError("Up: the skip tensor must be at least twice the size of x1")
            pt, pl = (H2 - 2 * H) // 2, (W2 - 2 * W) // 2
            Ct = Cs + Cu
            alloc = torch.zeros if (H2 != 2 * H or W2 != 2 * W) else torch.empty
            cat = alloc((N, H2, W2, Ct), dtype=tdt, device=dev)
            _to_nhwc(skip, cat, Ct, 0)                                 # torch.cat([x2, x1], dim=1): skip first
            x16 = _to_nhwc(x)
            if bilinear:
                ops.upsample2x_bilinear_fwd(x16, cat, N, H, W, Cx, H2, W2, out_stride=Ct, out_coff=Cs, ooy=pt, oox=pl)
            else:
                wf = torch.empty((4, Cu, Cx), dtype=tdt, device=dev)
                wd = torch.empty((4, Cx, Cu), dtype=tdt, device=dev)
                ops.pack_weight(module.up.weight.detach().contiguous(), wf, wd, True)
                ops.upconv2x2_fwd(x16, wf, module.up.bias.detach(), cat, N, 1, H, W, Cx, Cu, 1, H2, W2, out_stride=Ct,
                                  out_coff=Cs, ooy=pt, oox=pl)
                state["up_wd"] = wd
            state.update(x16=x16, bilinear=bilinear, Cs=Cs, Cu=Cu, Ct=Ct, pt=pt, pl=pl, H2=H2, W2=W2, skip_shape=tuple(skip.shape))
            inp, image, in_stride, h, w = cat, False, None, H2, W2
        cmid = s0.conv.weight.shape[0]
        cout = s3.conv.weight.shape[0]
        if cmid % 8 or cout % 8:
            raise NotImplementedError("stand-alone blocks: mid / out channels must be multiples of 8")
        zmid = torch.empty((N, h, w, cmid), dtype=tdt, device=dev)
        s0.forward(inp, image, N, h, w, training, zmid, cmid, 0, in_stride)
        zout = torch.empty((N, h, w, cout), dtype=tdt, device=dev)
        s3.forward(zmid, False, N, h, w, training, zout, cout, 0)
        ctx.state, ctx.module = state, module
        ctx.need_dx = x.requires_grad
        ctx.need_dskip = skip is not None and skip.requires_grad
        return _to_nchw(zout, cout)

    @staticmethod
    def backward(ctx, dout):
        st, module = ctx.state, ctx.module
        s0, s3, kind = st["s0"], st["s3"], st["kind"]
        tdt = _tdt()
        r = _grad_scale(dout)
        inv_r = torch.reciprocal(r)
        dz = _to_nhwc(dout.contiguous().float() * r)
        dmid, dw3, dg3, db3 = s3.backward(dz, s3.cout, 0, True)
        need_dinp = kind != "double" or ctx.need_dx
        dinp, dw0, dg0, db0 = s0.backward(dmid, s0.cout, 0, need_dinp)
        grads = [dw0, dg0, db0, dw3, dg3, db3]
        dx = dskip = None
        N = s0.N
        if kind == "double":
            if ctx.need_dx:
                dx = dinp if s0.image else _to_nchw(dinp, s0.cin)
        elif kind == "down":
            x16 = st["x16"]
            _, H, W, C = x16.shape
            dx16 = torch.empty_like(x16)
            # max-pool backward: the gradient of the pooled tensor is routed to the first maximum of every 2x2 window
            ops.bn_act_bwd_apply(x16, None, C, 0, dinp, None, None, None, None, None, None, ACT_NONE, False, dx16)
            dx = _to_nchw(dx16, C)
        else:
            Cs, Cu, Ct, pt, pl, H2, W2 = st["Cs"], st["Cu"], st["Ct"], st["pt"], st["pl"], st["H2"], st["W2"]
            x16 = st["x16"]
            _, h, w, Cx = x16.shape
            if ctx.need_dskip:
                dskip = _to_nchw(dinp, Cs, Ct, 0)
            dx16 = torch.empty_like(x16)
            if st["bilinear"]:
                ops.upsample2x_bilinear_bwd(dinp, dx16, N, h, w, Cx, H2, W2, dy_stride=Ct, dy_coff=Cs, ooy=pt, oox=pl)
            else:
                taps = [(py + pt, px + pl) for py in range(2) for px in range(2)]
                g = ops.make_geom(N, H2, W2, Cu, h, w, Cx, h, w, taps, isy=2, isx=2, in_stride=Ct, in_coff=Cs)
                db = torch.empty(Cu, dtype=torch.float32, device=dout.device)
                col_ws = torch.empty(1024 * Cu, dtype=torch.float32, device=dout.device)
                ops.colsum(dinp, Ct, Cs, N, H2, W2, pt, pl, 2 * h, 2 * w, Cu, 1.0, col_ws, db)
                dwu = torch.empty_like(module.up.weight, memory_format=torch.contiguous_format)
                ws = torch.empty(ops.conv_wgrad_ws_floats(g), dtype=torch.float32, device=dout.device)
                ops.conv_wgrad_det(g, dinp, x16, ws, dwu, Cx, Cu, 4, 1.0)
                ops.conv_igemm(g, dinp, st["up_wd"], dx16)
                grads += [dwu, db]
            dx = _to_nchw(dx16, Cx)
        torch._foreach_mul_(grads, inv_r)
        if dx is not None:
            dx = dx * inv_r
        if dskip is not None:
            dskip = dskip * inv_r
        return (None, None, None, dx if ctx.need_dx else None, dskip, *grads)


def double_conv_forward(module, x):
    _check(x, "DoubleConv")
    return _BlockFunction.apply("double", module, module.training, x, None, *_dc_params(module))


def down_forward(module, x):
    _check(x, "Down")
    return _BlockFunction.apply("down", module, module.training, x, None, *_dc_params(module.maxpool_conv[1]))


def up_forward(module, x1, x2):
    _check(x1, "Up")
    _check(x2, "Up")
    params = _dc_params(module.conv)
    if isinstance(module.up, torch.nn.ConvTranspose2d):
        params = params + [module.up.weight, module.up.bias]
    return _BlockFunction.apply("up", module, module.training, x1, x2, *params)


class _OutConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, weight, bias):
        N, C, H, W = x.shape
        x16 = _to_nhwc(x)
        logits = torch.empty((N, weight.shape[0], H, W), dtype=torch.float32, device=x.device)
        ops.conv_smallcout_fwd(x16, weight.detach().contiguous(), bias.detach() if bias is not None else None, logits)
        ctx.save_for_backward(x16, weight)
        ctx.has_bias, ctx.need_dx, ctx.C = bias is not None, x.requires_grad, C
        return logits

    @staticmethod
    def backward(ctx, dl):
        x16, weight = ctx.saved_tensors
        r = _grad_scale(dl)
        inv_r = torch.reciprocal(r)
        dw = torch.zeros_like(weight, memory_format=torch.contiguous_format)
        db = torch.zeros(weight.shape[0], dtype=torch.float32, device=dl.device) if ctx.has_bias else None
        dx16 = torch.empty_like(x16) if ctx.need_dx else None
        ops.conv_smallcout_bwd(x16, weight.detach().contiguous(), (dl.contiguous().float() * r), dx16, dw, db, gscale=1.0)
        dw = dw * inv_r
        if db is not None:
            db = db * inv_r
        dx = _to_nchw(dx16, ctx.C) * inv_r if ctx.need_dx else None
        return None, dx, dw, db


def out_conv_forward(module, x):
    _check(x, "OutConv")
    conv = module.conv
    C = x.shape[1]
    nch = C // 8
    if C % 8 or nch & (nch - 1) or conv.weight.shape[0] > 4 or conv.kernel_size != (1, 1):
        raise NotImplementedError("stand-alone OutConv: in_channels = 8 * 2^j, at most 4 classes, 1x1 kernel")
    return _OutConvFunction.apply(module, x, conv.weight, conv.bias)
