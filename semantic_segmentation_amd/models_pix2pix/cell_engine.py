"""Stand-alone execution of the mixed transposed-conv primitives (reference: architecture_pix2pix/operations.py:14-39
`re_conv_421/622/823(x)`; models_pix2pix/networks.py:486-511 `MixedOp_upconv(x, weights)`, `Cell_upconv(x)`).

Inside `UnetGenerator.forward` the cells run fused with their neighbours (pix2pix_engine.py).  Called on their own they take
and return fp32 NCHW tensors like the reference and run the same kernels: the weighted sum of the three transposed convs is
ONE merged 8x8 / stride-2 transposed conv (exact by linearity; a single primitive is the one-hot case), four sub-pixel
classes in one MFMA launch; backward = class weight gradients in one launch, split back into dW4 / dW6 / dW8 and the three
mixing-weight derivatives, and a 64-tap stride-2 data gradient.  First-order autograd; CPU tensors raise."""
from __future__ import annotations

import torch

from .. import ops
from .._lib import ACT_NONE
from ..unet.block_engine import _grad_scale, _tdt, _to_nchw, _to_nhwc

TAPS64 = [(ky - 3, kx - 3) for ky in range(8) for kx in range(8)]


class _MixedUpconvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weights3, w4, w6, w8, b4, b6, b8):
        if not x.is_cuda:
            raise RuntimeError("mixed transposed conv (semantic_segmentation_amd) runs on the MI355X only: no CPU / ATen fallback")
        N, Cin, H, W = x.shape
        Cout = w8.shape[1]
        if Cin % 8:
            raise NotImplementedError("stand-alone mixed transposed conv: C_in must be a multiple of 8")
        dev, tdt = x.device, _tdt()
        cpad = (Cout + 7) // 8 * 8
        sm = weights3.detach().float().contiguous().to(dev)

        def pad(w):
            if cpad == Cout:
                return w.detach().contiguous()
            z = torch.zeros((Cin, cpad, w.shape[2], w.shape[3]), dtype=torch.float32, device=dev)
            z[:, :Cout] = w.detach()
            return z
        w4p, w6p, w8p = pad(w4), pad(w6), pad(w8)
        pf = torch.empty((4, 16, cpad, Cin), dtype=tdt, device=dev)
        pd = torch.empty((64, Cin, cpad), dtype=tdt, device=dev)
        ops.upconv_merge_pack(w4p, w6p, w8p, sm, pf, None, None)
        ops.upconv_merge_pack(w4p, w6p, w8p, sm, None, pd, None)
        bias = None
        if b8 is not None:
            bias = torch.zeros(cpad, dtype=torch.float32, device=dev)
            bias[:Cout] = sm[0] * b4.detach() + sm[1] * b6.detach() + sm[2] * b8.detach()
        x16 = _to_nhwc(x)
        geoms = []
        for cls in range(4):
            g = ops.geom_convT_class(N, H, W, Cin, cpad, 8, 3, cls >> 1, cls & 1)
            for t in range(g.ntaps):
                g.tap_w[t] = t
            geoms.append(g)
        u = torch.empty((N, 2 * H, 2 * W, cpad), dtype=tdt, device=dev)
        ops.conv_igemm_batch(geoms, x16, [pf[c] for c in range(4)], u, bias, None)
        ctx.save_for_backward(x16, sm, w4, w6, w8, b4, b6, b8)
        ctx.geoms, ctx.pd, ctx.dims = geoms, pd, (N, Cin, H, W, Cout, cpad)
        ctx.need_dx = x.requires_grad
        return _to_nchw(u, Cout, cpad, 0)

    @staticmethod
    def backward(ctx, dy):
        x16, sm, w4, w6, w8, b4, b6, b8 = ctx.saved_tensors
        N, Cin, H, W, Cout, cpad = ctx.dims
        dev, tdt = dy.device, _tdt()
        r = _grad_scale(dy)
        inv_r = torch.reciprocal(r)
        alloc = torch.zeros if cpad != Cout else torch.empty
        du = alloc((N, 2 * H, 2 * W, cpad), dtype=tdt, device=dev)
        ops.nchw_to_nhwc((dy.contiguous().float() * r).contiguous(), du, cpad, 0)
        dwm = torch.empty((4, 16, cpad, Cin), dtype=torch.float32, device=dev)
        nparts = ops.conv_wgrad_parts(ctx.geoms[0])
        wsl = torch.empty(nparts * dwm.numel(), dtype=torch.float32, device=dev) if nparts > 1 else None
        ops.conv_wgrad_det_batch(ctx.geoms, x16, du, wsl, dwm)
        if cpad != Cout:
            dwm = dwm[:, :, :Cout, :].contiguous()
        dw4 = torch.empty_like(w4, memory_format=torch.contiguous_format)
        dw6 = torch.empty_like(w6, memory_format=torch.contiguous_format)
        dw8 = torch.empty_like(w8, memory_format=torch.contiguous_format)
        dots = torch.zeros(3, dtype=torch.float32, device=dev)
        ops.upconv_split_wgrad(dwm, w4.detach().contiguous(), w6.detach().contiguous(), w8.detach().contiguous(), sm, 1.0,
                               dw4, dw6, dw8, dots)
        db = [None, None, None]
        if b8 is not None:
            colws = torch.empty(1024 * cpad, dtype=torch.float32, device=dev)
            dbm = torch.empty(cpad, dtype=torch.float32, device=dev)
            ops.colsum(du, cpad, 0, N, 2 * H, 2 * W, 0, 0, 2 * H, 2 * W, cpad, 1.0, colws, dbm)
            dbm = dbm[:Cout]
            for j, bj in enumerate((b4, b6, b8)):
                db[j] = (sm[j] * dbm * inv_r).contiguous()
                dots[j] += (dbm * bj.detach()).sum()
        dx = None
        if ctx.need_dx:
            dx16 = torch.empty((N, H, W, Cin), dtype=tdt, device=dev)
            g = ops.make_geom(N, 2 * H, 2 * W, cpad, H, W, Cin, H, W, TAPS64, isy=2, isx=2)
            ops.conv_igemm(g, du, ctx.pd, dx16)
            dx = _to_nchw(dx16, Cin) * inv_r
        return dx, dots * inv_r, dw4 * inv_r, dw6 * inv_r, dw8 * inv_r, db[0], db[1], db[2]


def mixed_upconv(x, weights3, ops_list):
    """sum_j weights3[j] * ops_list[j](x) for the (k4, k6, k8) transposed-conv primitives: MixedOp_upconv.forward."""
    w = [o.op.weight for o in ops_list]
    b = [o.op.bias for o in ops_list]
    if any(bi is None for bi in b):
        b = [None, None, None]
    return _MixedUpconvFunction.apply(x, weights3, w[0], w[1], w[2], b[0], b[1], b[2])


def single_upconv(module, x):
    """One primitive on its own (re_conv_421 / 622 / 823): the one-hot case of the merged kernel; the two absent kernels are
    zero tensors whose (zero-weighted) gradients are dropped."""
    k = module.op.kernel_size[0]
    j = {4: 0, 6: 1, 8: 2}[k]
    cin, cout = module.op.weight.shape[0], module.op.weight.shape[1]
    dev = x.device
    ws, bs = [], []
    for jj, kk in enumerate((4, 6, 8)):
        if jj == j:
            ws.append(module.op.weight)
            bs.append(module.op.bias)
        else:
            ws.append(torch.zeros((cin, cout, kk, kk), dtype=torch.float32, device=dev))
            bs.append(torch.zeros(cout, dtype=torch.float32, device=dev) if module.op.bias is not None else None)
    onehot = torch.zeros(3, dtype=torch.float32, device=dev)
    onehot[j] = 1.0
    return _MixedUpconvFunction.apply(x, onehot, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2])
