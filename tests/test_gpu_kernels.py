"""Per-kernel parity tests: every C-ABI entry point against a plain fp32 torch (CPU) statement of the
same op, on the same 16-bit-rounded inputs.  GPU only (`-m gpu`)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [("f16", torch.float16), ("bf16", torch.bfloat16)]


def dev():
    return torch.device("cuda:0")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / max(b.norm().item(), 1e-30))


def rnd(g, *shape, dt=None, scale=1.0):
    t = torch.randn(*shape, generator=g) * scale
    return t.to(dt).float() if dt is not None else t


def nhwc(t16_src: torch.Tensor, dt) -> torch.Tensor:
    """fp32 NCHW (already representable) -> device NHWC tensor of dtype dt"""
    return t16_src.permute(0, 2, 3, 1).contiguous().to(dt).to(dev())


def from_nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def tol(dt):
    return 3e-3 if dt == torch.float16 else 1.5e-2


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 13, 9, 64, 64), (1, 16, 16, 128, 192), (3, 7, 20, 72, 40),
                                            (1, 32, 32, 256, 128), (2, 21, 17, 8, 128), (1, 5, 6, 8, 72)])
def test_conv3x3_fwd_bn_partials(dtn, dt, N, H, W, Cin, Cout):
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(1)
    x = rnd(g, N, Cin, H, W, dt=dt)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05)
    ref = F.conv2d(x, w, None, padding=1)
    xd = nhwc(x, dt)
    wf = torch.empty(9, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(9, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.to(dev()), wf, wd, False)
    # pack layout check
    assert torch.equal(wf.float().cpu(), w.permute(2, 3, 0, 1).reshape(9, Cout, Cin))
    assert torch.equal(wd.float().cpu(), w.permute(2, 3, 1, 0).reshape(9, Cin, Cout))
    geom = ops.geom_conv(N, H, W, Cin, Cout, 3, 1, 1)
    mt = ops.conv_igemm_mtiles(geom)
    part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev())
    y = torch.full((N, H, W, Cout), float("nan"), dtype=dt, device=dev())
    ops.conv_igemm(geom, xd, wf, y, None, part)
    torch.cuda.synchronize()
    got = from_nhwc(y)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < tol(dt), rel_err(got, ref)
    p = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()
    s1, s2 = ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))
    assert (p[0] - s1).abs().max() < 1e-3 * max(1.0, s1.abs().max().item())
    assert (p[1] - s2).abs().max() < 1e-3 * s2.abs().max().item()


@pytest.mark.parametrize("dtn,dt", DTS)
def test_conv_strided_io_bias_act(dtn, dt):
    """channel-sliced input/output (concat buffers), bias and LeakyReLU epilogue, stride-2 4x4 conv"""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_LEAKY02
    g = torch.Generator().manual_seed(2)
    N, H, W, Cin, Cout = 2, 12, 10, 64, 96
    x = rnd(g, N, Cin, H, W, dt=dt)
    w = rnd(g, Cout, Cin, 4, 4, dt=dt, scale=0.05)
    b = rnd(g, Cout)
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.2)
    big = torch.zeros(N, H, W, Cin + 16, dtype=dt, device=dev())
    big[..., 8:8 + Cin] = nhwc(x, dt)
    wf = torch.empty(16, Cout, Cin, dtype=dt, device=dev())
    ops.pack_weight(w.to(dev()), wf, None, False)
    OH, OW = ref.shape[2], ref.shape[3]
    out = torch.zeros(N, OH, OW, Cout + 32, dtype=dt, device=dev())
    geom = ops.geom_conv(N, H, W, Cin, Cout, 4, 2, 1, in_stride=Cin + 16, in_coff=8, out_stride=Cout + 32, out_coff=32)
    ops.conv_igemm(geom, big, wf, out, b.to(dev()), None, ACT_LEAKY02)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out[..., 32:]), ref) < tol(dt)
    assert float(out[..., :32].abs().max()) == 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 13, 9, 64, 64), (1, 16, 16, 128, 192), (2, 8, 8, 256, 64),
                                            (3, 9, 5, 72, 40)])
def test_conv3x3_dgrad_wgrad(dtn, dt, N, H, W, Cin, Cout):
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(3)
    x = rnd(g, N, Cin, H, W, dt=dt).requires_grad_(True)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05).requires_grad_(True)
    dy = rnd(g, N, Cout, H, W, dt=dt)
    F.conv2d(x, w, None, padding=1).backward(dy)
    wf = torch.empty(9, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(9, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.detach().to(dev()), wf, wd, False)
    dyd = nhwc(dy, dt)
    dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev())
    ops.conv_igemm(ops.geom_conv_dgrad_s1(N, H, W, Cin, Cout, 3, 1), dyd, wd, dx)
    assert rel_err(from_nhwc(dx), x.grad) < tol(dt)
    dwp = torch.zeros(9, Cout, Cin, dtype=torch.float32, device=dev())
    ops.conv_wgrad(ops.geom_conv(N, H, W, Cin, Cout, 3, 1, 1), nhwc(x.detach(), dt), dyd, dwp)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev())
    ops.unpack_wgrad(dwp, dw, Cout, Cin, 9, False, 0.5)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu() * 2, w.grad) < 2e-3, rel_err(dw.cpu() * 2, w.grad)


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 32, 32, 128, 64), (2, 160, 160, 128, 64), (1, 16, 16, 64, 64)])
def test_dgrad_partials_give_upconv_bias_gradient(dtn, dt, N, H, W, Cin, Cout):
    """ConvTranspose2d bias gradient (unet_parts.py:53) = column sums of the second half of d(concat): taken from the tile
    partials of the data-gradient convolution that writes d(concat) (few tiles: one launch; many: two-stage)."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(31)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05)
    dy = rnd(g, N, Cout, H, W, dt=dt)
    ref = F.conv_transpose2d(dy.float(), w.float(), None, padding=1)            # = data gradient of conv2d(pad 1)
    wd = torch.empty(9, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.to(dev()), None, wd, False)
    mt = ops.conv3x3_stat_rows(N, H, W, Cout, Cin)              # rows the data-gradient launch (Cout -> Cin) writes
    part = torch.full((ops.bn_partials_numel(mt, Cin),), float("nan"), dtype=torch.float32, device=dev())
    dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev())
    ops.conv3x3(nhwc(dy, dt), wd, dx, N, H, W, Cout, Cin, ops.TAPS3_DGRAD, bn_partials=part)
    half = Cin // 2
    db = torch.full((half,), float("nan"), dtype=torch.float32, device=dev())
    ops.bn_partials_colsum(part, mt, Cin, half, half, 0.5, db)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), ref) < tol(dt)
    want = ref.double().sum((0, 2, 3))[half:]
    assert (db.double().cpu() * 2 - want).abs().max() < 2e-3 * max(1.0, want.abs().max().item())
    # and it agrees with the column-sum pass over the stored 16-bit tensor up to that tensor's rounding
    ws = torch.empty(1024 * half, dtype=torch.float32, device=dev())
    db2 = torch.empty(half, dtype=torch.float32, device=dev())
    ops.colsum(dx, Cin, half, N, H, W, 0, 0, H, W, half, 0.5, ws, db2)
    torch.cuda.synchronize()
    assert (db - db2).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtn,dt", DTS)
def test_wgrad_large_k_split(dtn, dt):
    """many pixels, 64 channels: exercises split-K + atomics and the 1x4 wave arrangement"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(4)
    N, H, W, Cin, Cout = 2, 96, 80, 64, 64
    x = rnd(g, N, Cin, H, W, dt=dt).requires_grad_(False)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    dy = rnd(g, N, Cout, H, W, dt=dt, scale=0.1)
    F.conv2d(x, w, None, padding=1).backward(dy)
    dwp = torch.zeros(9, Cout, Cin, dtype=torch.float32, device=dev())
    ops.conv_wgrad(ops.geom_conv(N, H, W, Cin, Cout, 3, 1, 1), nhwc(x, dt), nhwc(dy, dt), dwp)
    torch.cuda.synchronize()
    got = dwp.cpu().view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
    assert rel_err(got, w.grad) < 2e-3, rel_err(got, w.grad)


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("pad", [(0, 0), (1, 1)])
def test_conv_transpose2x2_fwd_bwd(dtn, dt, pad):
    """ConvTranspose2d(k2,s2)+bias written into the second half of a concat buffer (unet_parts.py:53-67)"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(5)
    N, h, w_, Cin, Cout = 2, 5, 6, 128, 64
    H2, W2 = 2 * h + pad[0], 2 * w_ + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, Cin, h, w_, dt=dt).requires_grad_(True)
    wt = rnd(g, Cin, Cout, 2, 2, dt=dt, scale=0.05).requires_grad_(True)
    b = rnd(g, Cout).requires_grad_(True)
    up = F.conv_transpose2d(x, wt, b, stride=2)
    up_p = F.pad(up, [pl, pad[1] - pl, pt, pad[0] - pt])
    dcat = rnd(g, N, 2 * Cout, H2, W2, dt=dt)
    up_p.backward(dcat[:, Cout:])
    wf = torch.empty(4, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(4, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(wt.detach().to(dev()), wf, wd, True)
    cat = torch.zeros(N, H2, W2, 2 * Cout, dtype=dt, device=dev())
    xd = nhwc(x.detach(), dt)
    for py in range(2):
        for px in range(2):
            geom = ops.make_geom(N, h, w_, Cin, h, w_, Cout, H2, W2, [(0, 0)], osy=2, osx=2, ooy=py + pt, oox=px + pl,
                                 out_stride=2 * Cout, out_coff=Cout)
            ops.conv_igemm(geom, xd, wf[py * 2 + px], cat, b.detach().to(dev()), None)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(cat[..., Cout:]), up_p.detach()) < tol(dt)
    assert float(cat[..., :Cout].abs().max()) == 0.0
    # backward
    taps = [(py + pt, px + pl) for py in range(2) for px in range(2)]
    gb = ops.make_geom(N, H2, W2, Cout, h, w_, Cin, h, w_, taps, isy=2, isx=2, in_stride=2 * Cout, in_coff=Cout)
    dcd = nhwc(dcat, dt)
    dx = torch.empty(N, h, w_, Cin, dtype=dt, device=dev())
    ops.conv_igemm(gb, dcd, wd, dx)
    assert rel_err(from_nhwc(dx), x.grad) < tol(dt)
    dwp = torch.zeros(4, Cin, Cout, dtype=torch.float32, device=dev())
    ops.conv_wgrad(gb, dcd, xd, dwp)
    dw = torch.empty(Cin, Cout, 2, 2, dtype=torch.float32, device=dev())
    ops.unpack_wgrad(dwp, dw, Cin, Cout, 4, False, 1.0)
    assert rel_err(dw.cpu(), wt.grad) < 2e-3
    ws = torch.empty(1024 * Cout, dtype=torch.float32, device=dev())
    db = torch.empty(Cout, dtype=torch.float32, device=dev())
    ops.colsum(dcd, 2 * Cout, Cout, N, H2, W2, pt, pl, 2 * h, 2 * w_, Cout, 1.0, ws, db)
    assert rel_err(db.cpu(), b.grad) < 1e-4


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,w,Cin,Cout,pad", [(2, 5, 6, 128, 64, (0, 0)), (2, 5, 6, 128, 64, (1, 1)),
                                                 (1, 16, 16, 1024, 512, (0, 0)), (3, 7, 3, 64, 8, (2, 3))])
def test_upconv2x2_merged_fwd(dtn, dt, N, h, w, Cin, Cout, pad):
    """gs_upconv2x2_fwd: the four sub-pixel classes of ConvTranspose2d(k2,s2)+bias in ONE pointwise GEMM, written
    (with the F.pad offset of unet_parts.py:58-61) into the second half of a zeroed concat buffer."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(6)
    H2, W2 = 2 * h + pad[0], 2 * w + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, Cin, h, w, dt=dt)
    wt = rnd(g, Cin, Cout, 2, 2, dt=dt, scale=0.05)
    b = rnd(g, Cout)
    ref = F.pad(F.conv_transpose2d(x, wt, b, stride=2), [pl, pad[1] - pl, pt, pad[0] - pt])
    wf = torch.empty(4, Cout, Cin, dtype=dt, device=dev())
    ops.pack_weight(wt.to(dev()), wf, None, True)
    cat = torch.zeros(N, H2, W2, 2 * Cout, dtype=dt, device=dev())
    ops.upconv2x2_fwd(nhwc(x, dt), wf, b.to(dev()), cat, N, 1, h, w, Cin, Cout, 1, H2, W2, out_stride=2 * Cout,
                      out_coff=Cout, ooy=pt, oox=pl)
    torch.cuda.synchronize()
    got = from_nhwc(cat[..., Cout:])
    # the padded border is never written; inside, compare with the torch result
    inner = torch.zeros_like(ref, dtype=torch.bool)
    inner[:, :, pt:pt + 2 * h, pl:pl + 2 * w] = True
    assert rel_err(got[inner], ref[inner]) < tol(dt)
    assert float(got[~inner].abs().max() if (~inner).any() else 0.0) == 0.0
    assert float(cat[..., :Cout].abs().max()) == 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,w,C,pad", [(2, 5, 6, 64, (0, 0)), (1, 8, 8, 128, (1, 1)), (3, 1, 4, 8, (0, 2)), (2, 16, 13, 72, (3, 0))])
def test_upsample2x_bilinear_fwd_bwd(dtn, dt, N, h, w, C, pad):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=True) + F.pad into the second half of the concat buffer
    (bilinear=True U-Net, unet_parts.py:49-50,58-67) and its gradient, vs ATen on the same 16-bit-rounded inputs."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(8)
    H2, W2 = 2 * h + pad[0], 2 * w + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, C, h, w, dt=dt).requires_grad_(True)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.pad(up, [pl, pad[1] - pl, pt, pad[0] - pt])
    dcat = rnd(g, N, 2 * C, H2, W2, dt=dt)
    ref.backward(dcat[:, C:])
    cat = torch.zeros(N, H2, W2, 2 * C, dtype=dt, device=dev())
    ops.upsample2x_bilinear_fwd(nhwc(x.detach(), dt), cat, N, h, w, C, H2, W2, out_stride=2 * C, out_coff=C, ooy=pt, oox=pl)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(cat[..., C:]), ref.detach()) < tol(dt)
    assert float(cat[..., :C].abs().max()) == 0.0
    dx = torch.empty(N, h, w, C, dtype=dt, device=dev())
    ops.upsample2x_bilinear_bwd(nhwc(dcat, dt), dx, N, h, w, C, H2, W2, dy_stride=2 * C, dy_coff=C, ooy=pt, oox=pl)
    assert rel_err(from_nhwc(dx), x.grad) < tol(dt)


@pytest.mark.parametrize("dtn,dt", DTS)
def test_upconv2x2x2_merged_fwd_3d(dtn, dt):
    """ConvTranspose3d(k2,s2)+bias (GenSeg-3D/UNet3D/unet3d.py:68) as one GEMM with 8 sub-voxel classes."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(7)
    N, D, h, w, Cin, Cout = 2, 3, 4, 5, 64, 64
    x = rnd(g, N, Cin, D, h, w, dt=dt)
    wt = rnd(g, Cin, Cout, 2, 2, 2, dt=dt, scale=0.05)
    b = rnd(g, Cout)
    ref = F.conv_transpose3d(x, wt, b, stride=2)                       # [N,Cout,2D,2h,2w]
    # pack: slot (kz*2+ky)*2+kx of [8][Cout][Cin]
    wf = wt.permute(2, 3, 4, 1, 0).reshape(8, Cout, Cin).contiguous().to(dt).to(dev())
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dt).to(dev())       # [N,D,h,w,Cin]
    y = torch.zeros(N, 2 * D, 2 * h, 2 * w, Cout + 8, dtype=dt, device=dev())
    ops.upconv2x2_fwd(xd, wf, b.to(dev()), y, N, D, h, w, Cin, Cout, 2 * D, 2 * h, 2 * w, out_stride=Cout + 8, out_coff=8)
    torch.cuda.synchronize()
    got = y[..., 8:].float().cpu().permute(0, 4, 1, 2, 3)
    assert rel_err(got, ref) < tol(dt)
    assert float(y[..., :8].abs().max()) == 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,D,h,w,Cin,Cout", [(1, 4, 8, 8, 128, 64), (2, 2, 8, 16, 256, 128), (1, 8, 16, 16, 128, 128)])
def test_upconv2x2x2_dma_gemm_path_3d(dtn, dt, N, D, h, w, Cin, Cout):
    """The same ConvTranspose3d on the LDS-DMA pointwise GEMM (csrc/pwgemm.hip: power-of-two maps, Cin % 128 == 0, Cout % 64 == 0,
    N*D*h*w % 256 == 0): eight sub-voxel classes scattered into a channel slice of a wider, deeper concat buffer."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(11)
    x = rnd(g, N, Cin, D, h, w, dt=dt)
    wt = rnd(g, Cin, Cout, 2, 2, 2, dt=dt, scale=0.05)
    b = rnd(g, Cout)
    ref = F.conv_transpose3d(x, wt, b, stride=2)
    wf = wt.permute(2, 3, 4, 1, 0).reshape(8, Cout, Cin).contiguous().to(dt).to(dev())
    xin = torch.zeros(N, D, h, w, Cin + 64, dtype=dt, device=dev())
    xin[..., 64:] = x.permute(0, 2, 3, 4, 1).to(dt).to(dev())
    y = torch.full((N, 2 * D, 2 * h, 2 * w, Cout + 64), 3.0, dtype=dt, device=dev())
    ops.upconv2x2_fwd(xin, wf, b.to(dev()), y, N, D, h, w, Cin, Cout, 2 * D, 2 * h, 2 * w, in_stride=Cin + 64, in_coff=64,
                      out_stride=Cout + 64, out_coff=0)
    torch.cuda.synchronize()
    got = y[..., :Cout].float().cpu().permute(0, 4, 1, 2, 3)
    assert rel_err(got, ref) < tol(dt)
    assert bool((y[..., Cout:] == 3.0).all())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,C,pool", [(2, 8, 6, 64, True), (2, 9, 7, 64, True), (3, 5, 5, 128, False),
                                          (1, 4, 4, 1024, True), (2, 6, 6, 24, False)])
def test_bn_relu_pool_fwd_bwd(dtn, dt, N, H, W, C, pool):
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(6)
    y = rnd(g, N, C, H, W, dt=dt).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    z = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    zr = z.detach().to(dt).float()
    # fake conv-epilogue partials: per-tile sums over 2 tiles
    yd = nhwc(y.detach(), dt)
    flat = y.detach().permute(0, 2, 3, 1).reshape(-1, C).double()
    half = flat.shape[0] // 2
    ntiles = 2
    part = torch.zeros(ops.bn_partials_numel(ntiles, C), dtype=torch.float32, device=dev())
    pv = torch.stack([torch.stack([flat[:half].sum(0), (flat[:half] ** 2).sum(0)]),
                      torch.stack([flat[half:].sum(0), (flat[half:] ** 2).sum(0)])]).float()
    part[: ntiles * 2 * C] = pv.reshape(-1).to(dev())
    coef = torch.empty(4, C, dtype=torch.float32, device=dev())
    rmd, rvd = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    ops.bn_finalize(part, ntiles, C, N * H * W, gamma.detach().to(dev()), beta.detach().to(dev()), rmd, rvd, 0.1, 1e-5,
                    coef[0], coef[1], coef[2], coef[3])
    assert torch.allclose(rmd.cpu(), rm, atol=1e-5) and torch.allclose(rvd.cpu(), rv, rtol=1e-4, atol=1e-5)
    zbuf = torch.zeros(N, H, W, 2 * C, dtype=dt, device=dev())
    zp = torch.empty(N, H // 2, W // 2, C, dtype=dt, device=dev()) if pool else None
    ops.bn_act_apply(yd, coef[0], coef[1], ACT_RELU, zbuf, 2 * C, C, zp)
    torch.cuda.synchronize()
    got = from_nhwc(zbuf[..., C:])
    assert (got - z.detach()).abs().max() < (4e-3 if dt == torch.float16 else 3e-2)
    assert float(zbuf[..., :C].abs().max()) == 0.0
    if pool:
        assert torch.equal(from_nhwc(zp), F.max_pool2d(got, 2))
    # backward: dz from a concat-half (strided) + pooled gradient
    dza = rnd(g, N, C, H, W, dt=dt)
    loss = (z * dza).sum()
    dzp_t = None
    if pool:
        dzp_t = rnd(g, N, C, H // 2, W // 2, dt=dt)
        # pool the ROUNDED z like the kernel does (ties/arg-max follow the stored values)
        zq = z + (zr - z).detach()
        loss = loss + (F.max_pool2d(zq, 2) * dzp_t).sum()
    loss.backward()
    dbuf = torch.zeros(N, H, W, 2 * C, dtype=dt, device=dev())
    dbuf[..., C:] = nhwc(dza, dt)
    dzp = nhwc(dzp_t, dt) if pool else None
    nt = ops.bn_bwd_tiles_used(N, H, W, pool)
    assert nt <= ops.bn_bwd_tiles(N, H, W)
    part2 = torch.zeros(ops.bn_partials_numel(ops.bn_bwd_tiles(N, H, W), C), dtype=torch.float32, device=dev())
    ops.bn_act_bwd_reduce(yd, dbuf, 2 * C, C, dzp, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part2)
    dgamma = torch.empty(C, device=dev()); dbeta = torch.empty(C, device=dev())
    c12 = torch.empty(2, C, device=dev())
    ops.bn_bwd_coeffs(part2, nt, C, N * H * W, 0.25, dgamma, dbeta, c12[0], c12[1])
    dy = torch.empty(N, H, W, C, dtype=dt, device=dev())
    ops.bn_act_bwd_apply(yd, dbuf, 2 * C, C, dzp, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy)
    torch.cuda.synchronize()
    assert rel_err(dgamma.cpu() * 4, gamma.grad) < 5e-3, rel_err(dgamma.cpu() * 4, gamma.grad)
    assert rel_err(dbeta.cpu() * 4, beta.grad) < 5e-3
    assert rel_err(from_nhwc(dy), y.grad) < (5e-3 if dt == torch.float16 else 2e-2), rel_err(from_nhwc(dy), y.grad)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("Cin,k,s,p,bias", [(1, 3, 1, 1, False), (1, 3, 1, 1, True), (2, 4, 2, 1, True), (1, 4, 2, 1, False), (3, 3, 1, 1, False)])
@pytest.mark.parametrize("H,W", [(18, 22), (45, 53)])       # one ragged tile; several tiles with a ragged last one
def test_smallcin(dtn, dt, Cin, k, s, p, bias, H, W):
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(7)
    N, Cout = 2, 64
    x = torch.randn(N, Cin, H, W, generator=g).requires_grad_(True)
    w = (0.2 * torch.randn(Cout, Cin, k, k, generator=g)).requires_grad_(True)
    b = torch.randn(Cout, generator=g) if bias else None
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    y = torch.empty(N, OH, OW, Cout, dtype=dt, device=dev())
    mt = ops.conv_smallcin_mtiles(N, OH, OW)
    part = None if bias else torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev())
    ops.conv_smallcin_fwd(x.detach().to(dev()), w.detach().to(dev()), None if b is None else b.to(dev()), y, part, k, s, p)
    torch.cuda.synchronize()
    assert (from_nhwc(y) - ref.detach()).abs().max() < (5e-3 if dt == torch.float16 else 4e-2)
    if part is not None:
        pp = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()
        assert (pp[0] - ref.detach().double().sum((0, 2, 3))).abs().max() < 1e-2
        assert rel_err(pp[1], (ref.detach().double() ** 2).sum((0, 2, 3))) < 1e-4
    dy = rnd(g, N, Cout, OH, OW, dt=dt)
    ref.backward(dy)
    dyd = nhwc(dy, dt)
    dw = torch.zeros(Cout, Cin, k, k, dtype=torch.float32, device=dev())
    ops.conv_smallcin_wgrad(x.detach().to(dev()), dyd, dw, k, s, p, 0.5)
    dx = torch.empty(N, Cin, H, W, dtype=torch.float32, device=dev())
    ops.conv_smallcin_dgrad(dyd, w.detach().to(dev()), dx, k, s, p, 0.5)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu() * 2, w.grad) < 1e-4
    assert rel_err(dx.cpu() * 2, x.grad) < 1e-4


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("image", ["noise", "smooth"])
@pytest.mark.parametrize("N,H,W", [(2, 18, 22), (3, 45, 53), (2, 64, 96)])
def test_stem_without_conv_output_in_memory(dtn, dt, N, H, W, image):
    """Conv2d(1,64,3,p1) -> BatchNorm2d(train) -> ReLU (unet_parts.py:16-18) with the convolution output never stored:
    statistics from the image alone, fused forward, and the two backward passes re-forming y -- against torch autograd
    (fp64 for the statistics).  image = "smooth": 0.9 + 0.1 * low-pass noise -- positive, with a DC offset and strongly
    correlated taps like the real JSRT / ISIC inputs in [0,1]: the hard case for the closed-form w^T G w - mean^2 statistics
    (white noise has a tap Gram matrix close to n * I and never exercises the cancellation)."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(53)
    C = 64
    x = torch.randn(N, 1, H, W, generator=g)
    smooth = image == "smooth"
    if smooth:
        lp = F.avg_pool2d(F.pad(x, (4, 4, 4, 4), mode="reflect"), 9, stride=1)
        x = (0.9 + 0.1 * lp / lp.abs().max()).contiguous()
    w = (0.3 * torch.randn(C, 1, 3, 3, generator=g)).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    z = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    dz = rnd(g, N, C, H, W, dt=dt)
    z.backward(dz)
    xd, wd_ = x.to(dev()), w.detach().to(dev())
    # statistics
    mt = ops.conv_smallcin_mtiles(N, H, W)
    part = torch.full((ops.bn_partials_numel(mt, C),), float("nan"), dtype=torch.float32, device=dev())
    taps = torch.full((mt * 54,), float("nan"), dtype=torch.float32, device=dev())
    ops.stem_stats(xd, wd_, part, taps)
    coef = torch.empty(4, C, device=dev())
    rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    ops.bn_finalize(part, mt, C, N * H * W, gamma.detach().to(dev()), beta.detach().to(dev()), rm, rv, 0.1, 1e-5,
                    coef[0], coef[1], coef[2], coef[3])
    yd_ = F.conv2d(x.double(), w.detach().double(), None, padding=1)
    # (smooth: per-channel variance down to ~1e-4 of mean^2 -- fp32 sums of squares leave ~1e-4 relative on invstd)
    assert rel_err(coef[2], yd_.mean((0, 2, 3))) < 1e-4 + 1e-5
    assert rel_err(coef[3], 1 / torch.sqrt(yd_.var((0, 2, 3), unbiased=False) + 1e-5)) < (5e-4 if smooth else 1e-4)
    # forward
    zd = torch.full((N, H, W, C), float("nan"), dtype=dt, device=dev())
    ops.stem_fwd_bn(xd, wd_, coef[0], coef[1], ACT_RELU, zd)
    torch.cuda.synchronize()
    ztol = (4e-3 if dt == torch.float16 else 3e-2) * (3.0 if smooth else 1.0)
    assert (from_nhwc(zd) - z.detach()).abs().max() < ztol
    # pair form (mixed / precise plans): hi + lo carries the fp32 value, into a [hi | lo] buffer; the hi plane is what the
    # single-output form stores
    zpair = torch.full((N, H, W, 2 * C), float("nan"), dtype=dt, device=dev())
    ops.stem_fwd_bn_pair(xd, wd_, coef[0], coef[1], ACT_RELU, zpair)
    torch.cuda.synchronize()
    assert torch.equal(zpair[..., :C], zd)
    zsum = (zpair[..., :C].double() + zpair[..., C:].double()).permute(0, 3, 1, 2).cpu()
    zown = F.relu(yd_ * coef[0].double().cpu().view(1, C, 1, 1) + coef[1].double().cpu().view(1, C, 1, 1))    # with OUR coefficients
    assert (zsum - zown).abs().max() < (2e-5 if dt == torch.float16 else 3e-4) * max(1.0, float(zown.abs().max()))
    dzd = nhwc(dz, dt)
    nt = ops.stem_bwd_tiles(N, H, W)
    # ONE pass over z (its sign) and dz + closed-form BatchNorm / weight gradients (tap sums / Gram matrix of the image)
    s1p = torch.full((nt * 64,), float("nan"), dtype=torch.float32, device=dev())
    sws = torch.full((nt * 576,), float("nan"), dtype=torch.float32, device=dev())
    assert ops.stem_bwd_onepass(xd, zd, dzd, C, 0, ACT_RELU, s1p, sws)
    # the same pass reading z as the hi plane of the pair buffer (pixel stride 2C): identical sums
    s1q, swq = torch.full_like(s1p, float("nan")), torch.full_like(sws, float("nan"))
    assert ops.stem_bwd_onepass(xd, zpair, dzd, C, 0, ACT_RELU, s1q, swq, z_stride=2 * C)
    torch.cuda.synchronize()
    assert torch.equal(s1q, s1p) and torch.equal(swq, sws)
    dgamma3 = torch.full((C,), float("nan"), device=dev()); dbeta3 = torch.full((C,), float("nan"), device=dev())
    dw3 = torch.zeros(C, 1, 3, 3, device=dev())
    ops.stem_bwd_finalize(sws, s1p, taps, wd_, coef[0], coef[2], coef[3], True, 0.5, dw3, dgamma3, dbeta3, N, H, W)
    torch.cuda.synchronize()
    # references in fp64 on the same dz (the fp32 autograd reference itself is ~1e-3 off on the smooth images)
    xw, ww = x.double(), w.detach().double().requires_grad_(True)
    g64, b64 = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    F.relu(F.batch_norm(F.conv2d(xw, ww, None, padding=1), None, None, g64, b64, True, 0.1, 1e-5)).backward(dz.double())
    gtol = 2e-2 if smooth else 2e-3
    assert rel_err(dbeta3 * 2, b64.grad) < gtol and rel_err(dgamma3 * 2, g64.grad) < gtol, (rel_err(dbeta3 * 2, b64.grad), rel_err(dgamma3 * 2, g64.grad))
    assert rel_err(dw3.cpu() * 2, ww.grad) < (gtol if dt == torch.float16 else 5 * gtol), rel_err(dw3.cpu() * 2, ww.grad)
    # eval-mode statistics (constants): c1 = c2 = 0, the tap sums are not read
    dw4 = torch.zeros(C, 1, 3, 3, device=dev())
    ops.stem_bwd_finalize(sws, s1p, None, wd_, coef[0], coef[2], coef[3], False, 1.0, dw4, dgamma3, dbeta3, N, H, W)
    yl3 = y.detach().clone().requires_grad_(True)
    F.relu(yl3 * coef[0].cpu().view(1, C, 1, 1) + coef[1].cpu().view(1, C, 1, 1)).backward(dz)
    want4 = torch.nn.grad.conv2d_weight(x, w.shape, yl3.grad, padding=1)
    torch.cuda.synchronize()
    assert rel_err(dw4.cpu(), want4) < (2e-3 if dt == torch.float16 else 1e-2), rel_err(dw4.cpu(), want4)

@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,ncls", [(2, 18, 22, 2), (3, 45, 53, 1), (1, 64, 64, 4)])
def test_head1x1_on_conv_output_with_bn_relu(dtn, dt, N, H, W, ncls):
    """OutConv (unet_parts.py:74) reading the last stage's convolution output with BatchNorm + ReLU (unet_parts.py:20-21) on
    its load path: logits, weight and bias gradients against torch fp32 on the same stored y, and against the path through the
    stored activation."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(47)
    C = 64
    y16 = rnd(g, N, C, H, W, dt=dt)
    scale = 1 + 0.3 * torch.randn(C, generator=g); shift = 0.3 * torch.randn(C, generator=g)
    wh = (0.3 * torch.randn(ncls, C, 1, 1, generator=g)).requires_grad_(True)
    bh = torch.randn(ncls, generator=g).requires_grad_(True)
    z = F.relu(y16 * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1))
    ref = F.conv2d(z, wh, bh)
    dl = torch.randn(N, ncls, H, W, generator=g)
    ref.backward(dl)
    yd = nhwc(y16, dt)
    sc, sh = scale.to(dev()), shift.to(dev())
    whd, bhd, dld = wh.detach().to(dev()), bh.detach().to(dev()), dl.to(dev())
    logits = torch.full((N, ncls, H, W), float("nan"), dtype=torch.float32, device=dev())
    ops.head1x1_bn_fwd(yd, sc, sh, ACT_RELU, whd, bhd, logits)
    dw = torch.zeros(ncls, C, 1, 1, device=dev()); db = torch.zeros(ncls, device=dev())
    ops.head1x1_bn_wgrad(yd, sc, sh, ACT_RELU, whd, dld, dw, db, gscale=0.5)
    # through the stored 16-bit activation
    zd = torch.empty(N, H, W, C, dtype=dt, device=dev())
    ops.bn_act_apply(yd, sc, sh, ACT_RELU, zd, C, 0, None)
    logits2 = torch.empty_like(logits)
    ops.conv_smallcout_fwd(zd, whd, bhd, logits2)
    torch.cuda.synchronize()
    e1, e2 = rel_err(logits.cpu(), ref.detach()), rel_err(logits2.cpu(), ref.detach())
    assert e1 < 1e-5, e1                                  # fp32 activation: only the summation order differs
    assert e1 <= e2 + 1e-6, (e1, e2)
    assert rel_err(dw.cpu() * 2, wh.grad) < 1e-4 and rel_err(db.cpu() * 2, bh.grad) < 1e-4


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,C,ncls", [(2, 18, 22, 64, 2), (3, 45, 53, 64, 1), (2, 40, 40, 32, 4)])
def test_bn_bwd_head_source(dtn, dt, N, H, W, C, ncls):
    """BatchNorm/ReLU backward of the stage in front of the 1x1 head (unet_parts.py:19-21 then :74) with the gradient source
    formed from the logit gradient and the head weight: against a torch fp32 autograd reference, and against the path through
    the head's data-gradient tensor."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(43)
    y16 = rnd(g, N, C, H, W, dt=dt)
    yl = y16.clone().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    wh = 0.3 * torch.randn(ncls, C, 1, 1, generator=g)
    z = F.relu(F.batch_norm(yl, None, None, gamma, beta, True, 0.1, 1e-5))
    dl = torch.randn(N, ncls, H, W, generator=g)
    F.conv2d(z, wh).backward(dl)
    mean = y16.mean((0, 2, 3)); var = y16.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.detach() * invstd; shift = beta.detach() - mean * scale
    coef = torch.stack([scale, shift, mean, invstd]).to(dev()).contiguous()
    yd = nhwc(y16, dt)
    dld, whd = dl.to(dev()), wh.to(dev())
    part = torch.zeros(ops.bn_partials_numel(ops.bn_bwd_tiles(N, H, W), C), dtype=torch.float32, device=dev())
    nt = ops.bn_bwd_tiles_used(N, H, W, False)
    res = []
    for fused in (True, False):
        dgamma = torch.empty(C, device=dev()); dbeta = torch.empty(C, device=dev())
        c12 = torch.empty(2, C, device=dev())
        dy = torch.full((N, H, W, C), float("nan"), dtype=dt, device=dev())
        if fused:
            ops.bn_act_bwd_reduce_head(yd, dld, whd, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part)
            ops.bn_bwd_coeffs(part, nt, C, N * H * W, 1.0, dgamma, dbeta, c12[0], c12[1])
            ops.bn_act_bwd_apply_head(yd, dld, whd, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, dy)
        else:
            if C != 64:
                continue                    # the head's data-gradient kernel under test elsewhere; compare where it is the fast one
            dz = torch.empty(N, H, W, C, dtype=dt, device=dev())
            ops.conv_smallcout_bwd(None, whd, dld, dz, None, None)
            ops.bn_act_bwd_reduce(yd, dz, C, 0, None, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part)
            ops.bn_bwd_coeffs(part, nt, C, N * H * W, 1.0, dgamma, dbeta, c12[0], c12[1])
            ops.bn_act_bwd_apply(yd, dz, C, 0, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy)
        torch.cuda.synchronize()
        res.append((rel_err(dgamma, gamma.grad), rel_err(dbeta, beta.grad), rel_err(from_nhwc(dy), yl.grad)))
    t = 3e-3 if dt == torch.float16 else 2e-2
    assert res[0][0] < 5e-3 and res[0][1] < 5e-3 and res[0][2] < t, res
    if len(res) == 2:                       # fp32 gradient source: no worse than through the 16-bit tensor
        assert res[0][2] <= res[1][2] * 1.2 + 1e-5, res


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,sa,ca", [(2, 18, 22, 64, 0), (3, 45, 53, 128, 64), (2, 64, 96, 64, 0)])
def test_stem_bn_bwd_wgrad_fused(dtn, dt, N, H, W, sa, ca):
    """BatchNorm/ReLU backward apply + one-channel stem weight gradient in one pass (unet_parts.py:16-18, in_channels=1):
    against a torch fp32 autograd reference of conv -> BN(train) -> ReLU and against the two kernels it replaces."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(41)
    C = 64
    x = torch.randn(N, 1, H, W, generator=g)
    w = (0.3 * torch.randn(C, 1, 3, 3, generator=g)).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, padding=1)
    y16 = y_ref.detach().to(dt).float()                     # the stored convolution output
    yl = y16.clone().requires_grad_(True)
    z = F.relu(F.batch_norm(yl, None, None, gamma, beta, True, 0.1, 1e-5))
    dz = rnd(g, N, C, H, W, dt=dt)
    z.backward(dz)
    want_dy = yl.grad                                       # gradient w.r.t. the convolution output (fp32)
    want_dw = torch.nn.grad.conv2d_weight(x, w.shape, want_dy, padding=1)
    # device side: statistics of the stored y, coefficients, then the fused pass
    mean = y16.mean((0, 2, 3)); var = y16.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.detach() * invstd; shift = beta.detach() - mean * scale
    yd = nhwc(y16, dt)
    dzbuf = torch.zeros(N, H, W, sa, dtype=dt, device=dev())
    dzbuf[..., ca:ca + C] = nhwc(dz, dt)
    coef = torch.stack([scale, shift, mean, invstd]).to(dev()).contiguous()
    part = torch.zeros(ops.bn_partials_numel(ops.bn_bwd_tiles(N, H, W), C), dtype=torch.float32, device=dev())
    ops.bn_act_bwd_reduce(yd, dzbuf, sa, ca, None, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part)
    dgamma = torch.empty(C, device=dev()); dbeta = torch.empty(C, device=dev())
    c12 = torch.empty(2, C, device=dev())
    ops.bn_bwd_coeffs(part, ops.bn_bwd_tiles_used(N, H, W, False), C, N * H * W, 1.0, dgamma, dbeta, c12[0], c12[1])
    xd = x.to(dev())
    dw = torch.zeros(C, 1, 3, 3, dtype=torch.float32, device=dev())
    assert ops.stem_bn_bwd_wgrad(yd, dzbuf, sa, ca, xd, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, dw, 0.5)
    # the two kernels it replaces
    dy = torch.empty(N, H, W, C, dtype=dt, device=dev())
    ops.bn_act_bwd_apply(yd, dzbuf, sa, ca, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy)
    dw2 = torch.zeros(C, 1, 3, 3, dtype=torch.float32, device=dev())
    ops.conv_smallcin_wgrad(xd, dy, dw2, 3, 1, 1, 0.5)
    torch.cuda.synchronize()
    assert rel_err(dgamma, gamma.grad) < 5e-3 and rel_err(dbeta, beta.grad) < 5e-3
    e_fused, e_two = rel_err(dw.cpu() * 2, want_dw), rel_err(dw2.cpu() * 2, want_dw)
    assert e_fused < (2e-3 if dt == torch.float16 else 1e-2), e_fused
    assert e_fused <= e_two * 1.2 + 1e-5, (e_fused, e_two)          # fp32 dy in registers: no worse than via the 16-bit tensor
    # accumulation contract: dw += ...
    assert ops.stem_bn_bwd_wgrad(yd, dzbuf, sa, ca, xd, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, dw, 0.5)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), want_dw) < (2e-3 if dt == torch.float16 else 1e-2)


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("Cin,Cout,k,p", [(64, 2, 1, 0), (64, 1, 1, 0), (512, 1, 4, 1)])
def test_smallcout(dtn, dt, Cin, Cout, k, p):
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(8)
    N, H, W = 2, 11, 9
    x = rnd(g, N, Cin, H, W, dt=dt).requires_grad_(True)
    w = (0.1 * torch.randn(Cout, Cin, k, k, generator=g)).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    y = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=dev())
    xd = nhwc(x.detach(), dt)
    ops.conv_smallcout_fwd(xd, w.detach().to(dev()), b.detach().to(dev()), y, k, 1, p)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref.detach()) < 1e-5
    dy = torch.randn(N, Cout, OH, OW, generator=g)
    ref.backward(dy)
    dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev())
    dw = torch.zeros_like(w.detach()).to(dev()); db = torch.zeros(Cout, device=dev())
    ops.conv_smallcout_bwd(xd, w.detach().to(dev()), dy.to(dev()), dx, dw, db, k, 1, p, 0.5)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), x.grad) < tol(dt)
    assert rel_err(dw.cpu() * 2, w.grad) < 1e-4
    assert rel_err(db.cpu() * 2, b.grad) < 1e-4


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [1, 2, 3])
@pytest.mark.parametrize("mode", ["rand", "zeros", "ones"])
def test_seg_loss_matches_oracle(C, mode):
    from oracle import oracle
    from semantic_segmentation_amd.losses import seg_loss
    g = torch.Generator().manual_seed(9)
    N, H, W = 3, 37, 29
    logits = (2 * torch.randn(N, C, H, W, generator=g)).requires_grad_(True)
    hi = max(C, 2)
    mask = torch.randint(0, hi, (N, 1, H, W), generator=g)
    if mode == "zeros":
        mask.zero_()
    elif mode == "ones":
        mask.fill_(1 if C <= 2 else C - 1)
    ref = oracle.seg_loss(logits, mask)
    ref.backward()
    lg = logits.detach().to(dev()).requires_grad_(True)
    loss, parts = seg_loss(lg, mask.to(dev()), return_parts=True)
    (loss * 3.0).backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1, abs(ref.item())), (loss.item(), ref.item())
    assert rel_err(lg.grad.cpu() / 3.0, logits.grad) < 1e-4


def test_dice_api_matches_golden(golden_dir):
    import os
    from semantic_segmentation_amd.util.dice_score import dice_coeff, dice_loss, multiclass_dice_coeff
    z = np.load(os.path.join(golden_dir, "dice_cases.npz"))
    p, t = torch.from_numpy(z["p"]).to(dev()), torch.from_numpy(z["t"]).to(dev())
    assert abs(dice_coeff(p, t, False).item() - float(z["coeff_rbf_false"])) < 1e-6
    assert abs(dice_coeff(p, t, True).item() - float(z["coeff_rbf_true"])) < 1e-6
    assert abs(dice_loss(p, t).item() - float(z["loss"])) < 1e-6
    zz = torch.zeros_like(p)
    assert dice_loss(zz, zz).item() == 0.0
    assert abs(dice_loss(p, zz).item() - float(z["zero_target_loss"])) < 1e-6
    assert abs(dice_coeff(p[0], t[0]).item() - float(z["coeff_2d"])) < 1e-6
    pm, tm = torch.from_numpy(z["pm"]).to(dev()), torch.from_numpy(z["tm"]).to(dev())
    assert abs(multiclass_dice_coeff(pm, tm, False).item() - float(z["mc_rbf_false"])) < 1e-6
    assert abs(dice_loss(pm, tm, multiclass=True).item() - float(z["mc_loss"])) < 1e-6
    pg = p.clone().requires_grad_(True)
    dice_loss(pg, t).backward()
    np.testing.assert_allclose(pg.grad.cpu().numpy(), z["loss_grad_p"], rtol=1e-4, atol=1e-9)


def test_mean_losses_match_golden(golden_dir):
    import os
    from semantic_segmentation_amd import losses
    z = np.load(os.path.join(golden_dir, "ops_micro.npz"))
    pred = torch.from_numpy(z["gan/pred"]).to(dev())
    for mode, code in (("vanilla", losses.MODE_BCE_CONST), ("lsgan", losses.MODE_MSE_CONST)):
        assert abs(losses.mean_loss(pred, None, 1.0, code).item() - float(z[f"gan/{mode}/real"])) < 1e-6
        assert abs(losses.mean_loss(pred, None, 0.0, code).item() - float(z[f"gan/{mode}/fake"])) < 1e-6
    assert abs(losses.mean_loss(pred, None, -1.0, losses.MODE_MEAN).item() - float(z["gan/wgangp/real"])) < 1e-6
    assert abs(losses.mean_loss(pred, None, 1.0, losses.MODE_MEAN).item() - float(z["gan/wgangp/fake"])) < 1e-6
    a, b = torch.from_numpy(z["l1/a"]).to(dev()), torch.from_numpy(z["l1/b"]).to(dev())
    assert abs(losses.l1_loss(a, b).item() - float(z["l1/y"])) < 1e-6
    x, t = torch.from_numpy(z["bce/x"]).to(dev()), torch.from_numpy(z["bce/t"]).to(dev())
    assert abs(losses.bce_with_logits(x, t).item() - float(z["bce/y"])) < 1e-6
    # gradients vs torch autograd on CPU
    xc = torch.from_numpy(z["bce/x"]).requires_grad_(True)
    F.binary_cross_entropy_with_logits(xc, torch.from_numpy(z["bce/t"])).backward()
    xg = x.clone().requires_grad_(True)
    losses.bce_with_logits(xg, t).backward()
    assert rel_err(xg.grad.cpu(), xc.grad) < 1e-5
    ac = torch.from_numpy(z["l1/a"]).requires_grad_(True)
    F.l1_loss(ac, torch.from_numpy(z["l1/b"])).backward()
    ag = a.clone().requires_grad_(True)
    losses.l1_loss(ag, b).backward()
    assert rel_err(ag.grad.cpu(), ac.grad) < 1e-6


def test_layout_roundtrip():
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 5, 7, 6, generator=g).half().float()
    d = torch.zeros(2, 7, 6, 8, dtype=torch.float16, device=dev())
    ops.nchw_to_nhwc(x.to(dev()), d, 8, 3)
    back = torch.empty(2, 5, 7, 6, dtype=torch.float32, device=dev())
    ops.nhwc_to_nchw(d, back, 8, 3, 2.0)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x * 2)
    assert float(d[..., :3].abs().max()) == 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 13, 9, 64, 64), (1, 16, 16, 128, 192), (2, 37, 41, 64, 64),
                                            (1, 40, 33, 128, 128), (3, 7, 20, 72, 40), (1, 32, 64, 192, 64)])
def test_conv3x3_halo_fwd_dgrad(dtn, dt, N, H, W, Cin, Cout):
    """halo-reuse 3x3 kernel: forward with BN partials into a strided output, and the data gradient"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(12)
    x = rnd(g, N, Cin, H, W, dt=dt).requires_grad_(True)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05).requires_grad_(True)
    ref = F.conv2d(x, w, None, padding=1)
    dy = rnd(g, N, Cout, H, W, dt=dt)
    ref.backward(dy)
    wf = torch.empty(9, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(9, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.detach().to(dev()), wf, wd, False)
    mt = ops.conv3x3_mtiles(N, H, W, Cout)
    part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev())
    y = torch.zeros(N, H, W, Cout + 8, dtype=dt, device=dev())
    ops.conv3x3(nhwc(x.detach(), dt), wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part, out_stride=Cout + 8, out_coff=8)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y[..., 8:]), ref.detach()) < tol(dt)
    assert float(y[..., :8].abs().max()) == 0.0
    p = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()
    s1, s2 = ref.detach().double().sum((0, 2, 3)), (ref.detach().double() ** 2).sum((0, 2, 3))
    assert (p[0] - s1).abs().max() < 1e-3 * max(1.0, s1.abs().max().item())
    assert (p[1] - s2).abs().max() < 1e-3 * s2.abs().max().item()
    dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev())
    ops.conv3x3(nhwc(dy, dt), wd, dx, N, H, W, Cout, Cin, ops.TAPS3_DGRAD)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), x.grad) < tol(dt)


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 13, 9, 64, 64), (1, 16, 16, 128, 192), (2, 37, 41, 64, 64),
                                            (1, 40, 33, 128, 64), (3, 7, 20, 72, 40), (4, 64, 64, 64, 128)])
def test_conv3x3_halo_wgrad(dtn, dt, N, H, W, Cin, Cout):
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(13)
    x = rnd(g, N, Cin, H, W, dt=dt)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    dy = rnd(g, N, Cout, H, W, dt=dt, scale=0.1)
    F.conv2d(x, w, None, padding=1).backward(dy)
    # channel-sliced operands (concat buffers)
    xb = torch.zeros(N, H, W, Cin + 8, dtype=dt, device=dev()); xb[..., 8:] = nhwc(x, dt)
    db = torch.zeros(N, H, W, Cout + 16, dtype=dt, device=dev()); db[..., :Cout] = nhwc(dy, dt)
    dwp = torch.zeros(9, Cout, Cin, dtype=torch.float32, device=dev())
    ops.conv3x3_wgrad(xb, db, dwp, N, H, W, Cin, Cout, in_stride=Cin + 8, in_coff=8, out_stride=Cout + 16, out_coff=0)
    torch.cuda.synchronize()
    got = dwp.cpu().view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
    assert rel_err(got, w.grad) < 2e-3, rel_err(got, w.grad)
    # deterministic form (slabs + ordered reduction fused with scale / unpack): reference layout, bit-reproducible
    ws = torch.full((ops.conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout),), float("nan"), dtype=torch.float32, device=dev())
    g1 = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev())
    g2 = torch.empty_like(g1)
    for gg in (g1, g2):
        ops.conv3x3_wgrad_det(xb, db, ws, gg, N, H, W, Cin, Cout, 0.25, in_stride=Cin + 8, in_coff=8, out_stride=Cout + 16,
                              out_coff=0)
    assert torch.equal(g1, g2)
    assert rel_err(g1.cpu() * 4, w.grad) < 2e-3


@pytest.mark.parametrize("A,B,T,parts,transposed", [(128, 128, 9, 4, 0), (256, 128, 9, 32, 0), (160, 130, 9, 3, 0), (128, 128, 16, 5, 0),
                                                    (128, 256, 4, 2, 0), (512, 512, 9, 1, 0), (128, 128, 9, 40, 0), (128, 256, 4, 2, 1),
                                                    (64, 64, 9, 7, 0)])
def test_wgrad_reduce_unpack_layouts(A, B, T, parts, transposed):
    """gs_wgrad_reduce_unpack: slabs [part][tap][A][B] -> reference layout [A][B][tap] ([B][A][tap] transposed) * gscale, in the
    few-parts row kernel (deep layers) and the many-parts kernel; overwrite semantics, repeatable bit for bit."""
    from semantic_segmentation_amd import _lib
    from semantic_segmentation_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(17)
    ws = torch.randn(parts, T, A, B, generator=g).to(dev())
    want = ws.double().sum(0).permute(2, 1, 0) if transposed else ws.double().sum(0).permute(1, 2, 0)    # [B][A][T] / [A][B][T]
    outs = []
    for _ in range(2):
        grad = torch.full((A * B * T,), float("nan"), dtype=torch.float32, device=dev())
        _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), A, B, T, transposed, 0.5, _stream())
        torch.cuda.synchronize()
        outs.append(grad)
    assert torch.equal(outs[0], outs[1])
    got = outs[0].double().cpu().view(*want.shape) * 2
    assert (got - want.cpu()).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("NB,D,H,W,Cin,Cout", [(1, 4, 8, 8, 64, 64), (2, 3, 9, 13, 128, 72), (1, 1, 16, 16, 64, 8),
                                               (1, 5, 33, 20, 192, 64), (1, 4, 12, 12, 32, 64), (1, 3, 8, 8, 72, 32)])
def test_conv3d_3x3x3_halo_fwd_dgrad_wgrad(dtn, dt, NB, D, H, W, Cin, Cout):
    """gs_conv3d_3x3x3 / _wgrad: Conv3d(k3,p1) of GenSeg-3D/UNet3D/unet3d.py:28-31 forward, data gradient and weight
    gradient on the halo-reuse kernels ((depth tap, channel chunk) K stages), plus the BatchNorm partial sums."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(11)
    x = rnd(g, NB, Cin, D, H, W, dt=dt).requires_grad_(True)
    w = rnd(g, Cout, Cin, 3, 3, 3, dt=dt, scale=0.05).requires_grad_(True)
    y = F.conv3d(x, w, None, padding=1)
    dyt = rnd(g, NB, Cout, D, H, W, dt=dt)
    y.backward(dyt)

    def to_slices(t):                       # [NB,C,D,H,W] -> [NB*D,H,W,C]
        return t.permute(0, 2, 3, 4, 1).reshape(NB * D, H, W, t.shape[1]).contiguous().to(dt).to(dev())

    def from_slices(t, C):
        return t.float().cpu().reshape(NB, D, H, W, C).permute(0, 4, 1, 2, 3)

    wf = torch.empty(27, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(27, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.detach().reshape(Cout, Cin, 27, 1).to(dev()), wf, wd, False)
    xs, dys = to_slices(x.detach()), to_slices(dyt)
    nt = ops.conv3d3_mtiles(NB, D, H, W, Cout)
    part = torch.empty(ops.bn_partials_numel(nt, Cout), dtype=torch.float32, device=dev())
    yo = torch.empty(NB * D, H, W, Cout, dtype=dt, device=dev())
    ops.conv3d3(xs, wf, yo, NB, D, H, W, Cin, Cout, bn_partials=part)
    torch.cuda.synchronize()
    assert rel_err(from_slices(yo, Cout), y.detach()) < tol(dt)
    p = part[: nt * 2 * Cout].view(nt, 2, Cout).double().sum(0).cpu()
    ref1 = y.detach().double().sum(dim=(0, 2, 3, 4))
    ref2 = (y.detach().double() ** 2).sum(dim=(0, 2, 3, 4))
    assert float((p[0] - ref1).abs().max() / (ref1.abs().max() + 1e-6)) < 2e-3
    assert float((p[1] - ref2).abs().max() / (ref2.abs().max() + 1e-6)) < 2e-3
    if True:                                # data gradient: the same kernel with K = Cout (zero-padded tail chunk)
        dxo = torch.empty(NB * D, H, W, Cin, dtype=dt, device=dev())
        ops.conv3d3(dys, wd, dxo, NB, D, H, W, Cout, Cin, dgrad=True)
        assert rel_err(from_slices(dxo, Cin), x.grad) < tol(dt)
    dwp = torch.zeros(27, Cout, Cin, dtype=torch.float32, device=dev())
    ops.conv3d3_wgrad(xs, dys, dwp, NB, D, H, W, Cin, Cout)
    dw = torch.empty(Cout, Cin, 27, dtype=torch.float32, device=dev())
    ops.unpack_wgrad(dwp, dw, Cout, Cin, 27, False, 1.0)
    assert rel_err(dw.cpu().view(Cout, Cin, 3, 3, 3), w.grad) < 2e-3
    # deterministic form: slabs + ordered reduction straight into [Cout][Cin][27], twice -> bit-identical
    ws = torch.empty(ops.conv3d3_wgrad_ws_floats(NB, D, H, W, Cin, Cout), dtype=torch.float32, device=dev())
    d1 = torch.empty(Cout, Cin, 27, dtype=torch.float32, device=dev())
    d2 = torch.empty_like(d1)
    ops.conv3d3_wgrad_det(xs, dys, ws, d1, NB, D, H, W, Cin, Cout, 0.5)
    ws.fill_(float("nan"))                      # the workspace needs no initialisation
    ops.conv3d3_wgrad_det(xs, dys, ws, d2, NB, D, H, W, Cin, Cout, 0.5)
    assert torch.equal(d1, d2)
    assert rel_err(d1.cpu().view(Cout, Cin, 3, 3, 3) * 2, w.grad) < 2e-3


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,Cin,Cout,k,s,p", [(2, 2, 512, 512, 4, 2, 1), (2, 4, 512, 256, 4, 2, 1), (1, 8, 1024, 128, 3, 1, 1),
                                                 (2, 16, 512, 512, 4, 2, 1), (2, 12, 256, 384, 4, 2, 1), (3, 5, 192, 256, 3, 1, 1),
                                                 (2, 16, 320, 192, 4, 2, 1)])
def test_igemm_split_k_skinny(dtn, dt, N, h, Cin, Cout, k, s, p):
    """Skinny GEMMs (a few output pixels, K = taps x 512..1024 channels: the bottom of the Pix2Pix generator at batch 2,
    networks.py:582-602) go through the weight-streaming form (<= 128 output pixels, Cin % 64 == 0, Cout % 128 == 0:
    csrc/skinny.hip) or the split-K path of gs_conv_igemm: fp32 partial rows / tiles in the workspace, summed in part order
    before bias / activation / BatchNorm partial sums; the workspace needs no initialisation (NaN-filled between two calls)."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(13)
    x = rnd(g, N, Cin, h, h, dt=dt)
    w = rnd(g, Cout, Cin, k, k, dt=dt, scale=0.03)
    b = rnd(g, Cout)
    ref_raw = F.conv2d(x, w, None, stride=s, padding=p)
    ref = F.leaky_relu(ref_raw + b.view(1, -1, 1, 1), 0.2)
    oh = ref.shape[2]
    wf = torch.empty(k * k, Cout, Cin, dtype=dt, device=dev())
    ops.pack_weight(w.to(dev()), wf, None, False)
    geom = ops.geom_conv(N, h, h, Cin, Cout, k, s, p)
    from semantic_segmentation_amd._lib import ACT_LEAKY02
    outs = []
    for _ in range(2):
        y = torch.empty(N, oh, oh, Cout, dtype=dt, device=dev())
        ops.conv_igemm(geom, nhwc(x, dt), wf, y, b.to(dev()), None, ACT_LEAKY02)
        torch.cuda.synchronize()
        outs.append(from_nhwc(y))
        for ws in ops._SPLITK_WS.values():
            ws.fill_(float("nan"))
    assert rel_err(outs[0], ref) < tol(dt)
    assert torch.equal(outs[0], outs[1])
    # BatchNorm partial sums come from the full (summed) accumulators
    nt = ops.conv_igemm_mtiles(geom)
    part = torch.empty(ops.bn_partials_numel(nt, Cout), dtype=torch.float32, device=dev())
    y = torch.empty(N, oh, oh, Cout, dtype=dt, device=dev())
    ops.conv_igemm(geom, nhwc(x, dt), wf, y, None, part)
    torch.cuda.synchronize()
    ps = part[: nt * 2 * Cout].view(nt, 2, Cout).double().sum(0).cpu()
    r1 = ref_raw.double().sum(dim=(0, 2, 3))
    assert float((ps[0] - r1).abs().max() / (r1.abs().max() + 1e-6)) < 3e-3
    p2 = part[nt * 2 * Cout - Cout: nt * 2 * Cout].double().cpu() if nt == 1 else None
    if p2 is not None:
        r2 = (ref_raw.double() ** 2).sum(dim=(0, 2, 3))
        assert float((p2 - r2).abs().max() / (r2.abs().max() + 1e-6)) < 3e-3
    if os.environ.get("GSSEG_SPLITK", "1") != "0":
        assert ops._SPLITK_WS                                        # one workspace per (device, stream), passed per call


def test_igemm_weight_streaming_fuzz():
    """Random geometries of the weight-streaming igemm form (csrc/skinny.hip): conv k1..4 / stride 1..2 / pad 0..2, rectangular
    maps, strided input and output slices of wider buffers, bias + activation, BatchNorm partial rows -- against fp32 F.conv2d."""
    import random
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH
    rng = random.Random(20261004)
    g = torch.Generator().manual_seed(99)
    done = 0
    for trial in range(60):
        dt = rng.choice([torch.float16, torch.bfloat16])
        k = rng.choice([1, 2, 3, 4]); s = rng.choice([1, 2]); p = rng.choice([0, 1, 2])
        N = rng.choice([1, 2, 3]); IH = rng.randint(1, 24); IW = rng.randint(1, 24)
        Cin = 64 * rng.randint(1, 6); Cout = 32 * rng.randint(1, 10)
        OH, OW = (IH + 2 * p - k) // s + 1, (IW + 2 * p - k) // s + 1
        if OH < 1 or OW < 1 or p >= k:
            continue
        in_extra, out_extra = rng.choice([0, 64]), rng.choice([0, 32])
        geom = ops.geom_conv(N, IH, IW, Cin, Cout, k, s, p)
        geom.in_pix_stride, geom.in_coff = Cin + in_extra, rng.choice([0, in_extra])
        geom.out_pix_stride, geom.out_coff = Cout + out_extra, rng.choice([0, out_extra])
        M = N * OH * OW
        x = rnd(g, N, Cin, IH, IW, dt=dt)
        w = rnd(g, Cout, Cin, k, k, dt=dt, scale=0.05)
        b = rnd(g, Cout)
        act, actf = rng.choice([(ACT_NONE, lambda t: t), (ACT_RELU, torch.relu), (ACT_LEAKY02, lambda t: F.leaky_relu(t, 0.2)),
                                (ACT_TANH, torch.tanh)])
        raw = F.conv2d(x, w, None, stride=s, padding=p)
        ref = actf(raw + b.view(1, -1, 1, 1))
        wf = torch.empty(k * k, Cout, Cin, dtype=dt, device=dev())
        ops.pack_weight(w.to(dev()), wf, None, False)
        xin = torch.zeros(N, IH, IW, geom.in_pix_stride, dtype=dt, device=dev())
        xin[..., geom.in_coff:geom.in_coff + Cin] = nhwc(x, dt)
        y = torch.full((N, OH, OW, geom.out_pix_stride), 7.0, dtype=dt, device=dev())
        nt = ops.conv_igemm_mtiles(geom)
        part = torch.full((ops.bn_partials_numel(nt, Cout),), float("nan"), dtype=torch.float32, device=dev())
        ops.conv_igemm(geom, xin, wf, y, b.to(dev()), part, act)
        torch.cuda.synchronize()
        got = from_nhwc(y[..., geom.out_coff:geom.out_coff + Cout])
        assert rel_err(got, ref) < tol(dt), (trial, N, IH, IW, Cin, Cout, k, s, p)
        if geom.out_pix_stride > Cout:                      # the neighbouring channels of the wider buffer are untouched
            other = torch.ones(geom.out_pix_stride, dtype=torch.bool)
            other[geom.out_coff:geom.out_coff + Cout] = False
            assert bool((y[..., other.to(dev())] == 7.0).all()), trial
        ps = part[: nt * 2 * Cout].view(nt, 2, Cout).double().sum(0).cpu()
        r1, r2 = raw.double().sum(dim=(0, 2, 3)), (raw.double() ** 2).sum(dim=(0, 2, 3))
        assert float((ps[0] - r1).abs().max() / (r1.abs().max() + 1e-6)) < 5e-3, trial
        assert float((ps[1] - r2).abs().max() / (r2.abs().max() + 1e-6)) < 5e-3, trial
        done += int(nt == (M + 15) // 16)
    if os.environ.get("GSSEG_SKINNY", "1") != "0" and os.environ.get("GSSEG_SPLITK", "1") != "0":
        assert done >= 20, done                              # most trials ran on the weight-streaming form


@pytest.mark.parametrize("C", [1, 2])
def test_seg_loss_global_dice_equals_full_batch(C):
    """Data-parallel exact global-batch Dice: two 'ranks' each hold half of a batch; with the three Dice
    sums exchanged (emulated here by adding them) and the Dice-gradient multiplier = world, every rank reports the
    full-batch loss's Dice term and the AVERAGE of the rank gradients equals the full-batch gradient."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd.losses import _mask_u8, _ws, apply_global_dice, seg_loss
    g = torch.Generator().manual_seed(17)
    N, H, W = 4, 24, 20
    logits = torch.randn(N, C, H, W, generator=g).to(dev()).requires_grad_(True)
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.6).long().to(dev())
    loss_full, parts_full = seg_loss(logits, mask, return_parts=True)
    loss_full.backward()
    g_full = logits.grad.detach().clone()
    world, per = 2, N // 2
    outs, halves = [], []
    for r in range(world):
        lg = logits.detach()[r * per:(r + 1) * per].contiguous()
        mu = _mask_u8(mask[r * per:(r + 1) * per], per, H, W)
        out = torch.empty(8, dtype=torch.float32, device=dev())
        ops.seg_loss_fwd(lg, mu, _ws(dev()), out)
        outs.append(out)
        halves.append((lg, mu))
    sums = outs[0][3:6] + outs[1][3:6]                     # what dist.all_reduce(SUM) would produce
    grads = []
    for r in range(world):
        apply_global_dice(outs[r], sums, world)
        assert abs(float(outs[r][2]) - float(parts_full[2].detach())) < 1e-6   # global Dice loss on every rank
        d = torch.empty_like(halves[r][0])
        ops.seg_loss_bwd(halves[r][0], halves[r][1], outs[r], torch.ones(1, device=dev()), 1.0, d)
        grads.append(d)
    avg_equiv = torch.cat(grads, 0) / world                # gradient AVERAGE over ranks, sample by sample
    assert float((avg_equiv - g_full).abs().max()) < 1e-7 + 1e-5 * float(g_full.abs().max())
    ce_mean = 0.5 * (float(outs[0][1]) + float(outs[1][1]))
    assert abs(ce_mean - float(parts_full[1].detach())) < 1e-6


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,C,pad", [(2, 16, 24, 64, 0), (1, 17, 9, 72, 8), (3, 64, 64, 128, 128)])
def test_maxpool2x2_standalone_strided(dtn, dt, N, H, W, C, pad):
    """gs_maxpool2x2_fwd (unet_parts.py:34) reading the skip half of a concat buffer: bit-exact against F.max_pool2d."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(5)
    z = rnd(g, N, C, H, W, dt=dt)
    buf = torch.zeros(N, H, W, C + pad, dtype=dt, device=dev())
    buf[..., :C] = nhwc(z, dt)
    zp = torch.empty(N, H // 2, W // 2, C, dtype=dt, device=dev())
    ops.maxpool2x2_fwd(buf, zp, N, H, W, C, z_stride=C + pad, z_coff=0)
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(zp), F.max_pool2d(z, 2))


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 13, 9, 64, 64), (1, 40, 33, 128, 64), (2, 32, 32, 128, 256)])
def test_conv3x3_halo_bias_relu_epilogue(dtn, dt, N, H, W, Cin, Cout):
    """the inference form of a conv block: bias (folded BatchNorm) + ReLU in the halo kernel's epilogue, written into a
    channel slice of a wider buffer"""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(31)
    x = rnd(g, N, Cin, H, W, dt=dt)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05)
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    wf = torch.empty(9, Cout, Cin, dtype=dt, device=dev())
    ops.pack_weight(w.to(dev()), wf, None, False)
    y = torch.zeros(N, H, W, 2 * Cout, dtype=dt, device=dev())
    ops.conv3x3(nhwc(x, dt), wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, b.to(dev()), None, act=ACT_RELU,
                out_stride=2 * Cout, out_coff=Cout)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y[..., Cout:]), ref) < tol(dt)
    assert float(y[..., :Cout].abs().max()) == 0.0
    assert float(y.min()) >= 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,Cin,Cout,k,s,p", [(2, 4, 128, 256, 4, 2, 1), (2, 8, 512, 512, 4, 2, 1), (1, 3, 64, 72, 3, 1, 1)])
def test_conv_wgrad_assign_mode(dtn, dt, N, h, Cin, Cout, k, s, p):
    """gs_conv_wgrad_assign: few-pixel layers (no K split) write the weight gradient into an UN-zeroed buffer; a geometry
    that does split K is refused."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(8)
    x = rnd(g, N, Cin, h, h, dt=dt)
    w = rnd(g, Cout, Cin, k, k, dt=dt, scale=0.05).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    dy = rnd(g, *y.shape, dt=dt)
    y.backward(dy)
    geom = ops.geom_conv(N, h, h, Cin, Cout, k, s, p)
    assert ops.conv_wgrad_single_pass(geom)
    dwp = torch.full((k * k, Cout, Cin), float("nan"), dtype=torch.float32, device=dev())
    ops.conv_wgrad(geom, nhwc(x, dt), nhwc(dy, dt), dwp, assign=True)
    torch.cuda.synchronize()
    got = dwp.cpu().view(k, k, Cout, Cin).permute(2, 3, 0, 1)
    assert torch.isfinite(got).all()
    assert rel_err(got, w.grad) < tol(dt)
    big = ops.geom_conv(8, 64, 64, 64, 64, 3, 1, 1)
    assert not ops.conv_wgrad_single_pass(big)
    xb = torch.zeros(8, 64, 64, 64, dtype=dt, device=dev())
    with pytest.raises(RuntimeError):
        ops.conv_wgrad(big, xb, xb, torch.zeros(9, 64, 64, device=dev()), assign=True)


def test_jaccard_seg_loss_matches_golden_and_oracle(golden_dir):
    """BCE + Jaccard loss of the ISIC variant (train_end2end_isic.py:40-56,247-249): HIP kernel vs the reference-generated
    vectors (value + gradient) and vs the oracle on a larger random case with an empty mask."""
    import os
    from oracle import oracle
    from semantic_segmentation_amd.losses import seg_loss_jaccard
    z = np.load(os.path.join(golden_dir, "jaccard_cases.npz"))
    x = torch.from_numpy(z["logits"]).to(dev()).requires_grad_(True)
    m = torch.from_numpy(z["mask"].astype(np.int64)).to(dev())
    loss, parts = seg_loss_jaccard(x, m, return_parts=True)
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 2e-6
    assert abs(parts[1].item() - float(z["bce"])) < 2e-6
    assert np.abs(x.grad.cpu().numpy() - z["grad"]).max() < 2e-7
    assert abs(seg_loss_jaccard(x.detach()[:1], m[:1]).item() - float(z["loss_b1"])) < 2e-6
    g = torch.Generator().manual_seed(3)
    xl = (torch.randn(4, 1, 96, 128, generator=g) * 3)
    ml = (torch.rand(4, 1, 96, 128, generator=g) > 0.7).long()
    ml[2] = 0
    xr = xl.clone().requires_grad_(True)
    ref = oracle.seg_loss_jaccard(xr, ml)
    ref.backward()
    xd = xl.to(dev()).requires_grad_(True)
    got = seg_loss_jaccard(xd, ml.to(dev()))
    (got * 3.0).backward()
    assert abs(got.item() - ref.item()) < 2e-6
    assert (xd.grad.cpu() - 3.0 * xr.grad).abs().max() < 1e-8 + 2e-6 * xr.grad.abs().max()


@pytest.mark.parametrize("N,h,cin,cout", [(2, 4, 1024, 512), (2, 16, 512, 256), (8, 32, 256, 128), (2, 1, 512, 512),
                                          (2, 2, 1024, 512), (2, 8, 1024, 512)])
def test_igemm_batch_equals_per_class_launches(N, h, cin, cout):
    """gs_conv_igemm_batch (four sub-pixel classes of a merged k8/s2 transposed conv in one grid, incl. split-K through the
    workspace and BatchNorm partials) against four gs_conv_igemm launches (bit-identical on the register-staged engine) and
    against F.conv_transpose2d."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(h)
    dt = torch.float16
    x = torch.randn(N, h, h, cin, generator=g).to(dev()).to(dt)
    pf = (0.05 * torch.randn(4, 16, cout, cin, generator=g)).to(dev()).to(dt)
    bias = torch.randn(cout, generator=g).to(dev())
    geoms = []
    for cls in range(4):
        gm = ops.geom_convT_class(N, h, h, cin, cout, 8, 3, cls >> 1, cls & 1)
        for t in range(gm.ntaps):
            gm.tap_w[t] = t
        geoms.append(gm)
    mt = ops.conv_igemm_mtiles(geoms[0])
    npart = ops.bn_partials_numel(4 * mt, cout)
    y1 = torch.zeros(N, 2 * h, 2 * h, cout, dtype=dt, device=dev())
    p1 = torch.zeros(npart, device=dev())
    for cls in range(4):
        ops.conv_igemm(geoms[cls], x, pf[cls], y1, bias, p1[cls * mt * 2 * cout:])
    y2 = torch.zeros_like(y1)
    p2 = torch.zeros(npart, device=dev())
    ops.conv_igemm_batch(geoms, x, [pf[c] for c in range(4)], y2, bias, [p2[c * mt * 2 * cout:] for c in range(4)])
    torch.cuda.synchronize()
    if mt == (N * h * h + 127) // 128 and N * h * h > 16:
        assert torch.equal(y1, y2)
        assert torch.equal(p1[:4 * mt * 2 * cout], p2[:4 * mt * 2 * cout])
    else:
        # the weight-streaming form (csrc/skinny.hip) cuts K by the launch's block budget (one GEMM vs four): the same sums in
        # a different split
        assert float((y1.float() - y2.float()).abs().max()) <= 2e-3 * float(y1.float().abs().max())
        assert float((p1 - p2)[:4 * mt * 2 * cout].abs().max()) <= 1e-4 * float(p1.abs().max())
        y3 = torch.zeros_like(y1)
        ops.conv_igemm_batch(geoms, x, [pf[c] for c in range(4)], y3, bias, [p2[c * mt * 2 * cout:] for c in range(4)])
        assert torch.equal(y2, y3)                                   # run-to-run identical
    ref = torch.nn.functional.conv_transpose2d(x.float().permute(0, 3, 1, 2).cpu(),
                                               _unpack_merged(pf.float().cpu(), cin, cout), bias.cpu(), stride=2, padding=3)
    assert rel_err(from_nhwc(y2), ref) < tol(dt)


def _unpack_merged(pf, cin, cout):
    """class-major pack [4][16][Cout][Cin] -> ConvTranspose2d weight [Cin][Cout][8][8] (csrc/pix2pix.hip layout)."""
    w = torch.zeros(cin, cout, 8, 8)
    for ky in range(8):
        for kx in range(8):
            c = (1 - (ky & 1)) * 2 + (1 - (kx & 1))
            t = (ky >> 1) * 4 + (kx >> 1)
            w[:, :, ky, kx] = pf[c, t].t()
    return w


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("form", [0, 4, 8, 44])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 37, 41, 64, 64), (1, 17, 33, 128, 72), (3, 9, 100, 64, 128), (2, 48, 32, 192, 64),
                                            (2, 33, 40, 128, 256),
                                            # several items per block (the persistent grid is 256 blocks): the deferred epilogue's
                                            # hand-over between items, running statistics across items, resident weights (Cin = 64),
                                            # a ragged last row of patches in between
                                            (24, 96, 128, 64, 64), (10, 104, 96, 64, 128), (20, 64, 96, 128, 64),
                                            # ... with ragged patches on both edges (immediate epilogue between deferred ones, resident weights kept)
                                            (20, 100, 70, 64, 64), (12, 100, 70, 64, 136)])
def test_conv3x3_kernel_forms(dtn, dt, form, N, H, W, Cin, Cout):
    """every form of the 2-D kernel (register-staged big-K-step, LDS-DMA with 4 / 8 waves) on the
    same operands: forward with BatchNorm partials into a strided slice (ragged patches, a partial cout tile, an odd
    number of 8-row halves), bias + ReLU epilogue, and the data gradient (flipped tap table)"""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(31)
    x = rnd(g, N, Cin, H, W, dt=dt).requires_grad_(True)
    w = rnd(g, Cout, Cin, 3, 3, dt=dt, scale=0.05).requires_grad_(True)
    b = rnd(g, Cout)
    ref = F.conv2d(x, w, None, padding=1)
    dy = rnd(g, N, Cout, H, W, dt=dt)
    ref.backward(dy)
    wf = torch.empty(9, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(9, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(w.detach().to(dev()), wf, wd, False)
    mt = ops.conv3x3_mtiles(N, H, W, Cout)
    part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev())
    xin = torch.zeros(N, H, W, Cin + 16, dtype=dt, device=dev()); xin[..., 16:] = nhwc(x.detach(), dt)
    y = torch.zeros(N, H, W, Cout + 8, dtype=dt, device=dev())
    y2 = torch.empty(N, H, W, Cout, dtype=dt, device=dev())
    dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev())
    ops.conv3x3_set_kernel_form(form)
    try:
        for _ in range(2):          # the second launch runs on warm caches: same result
            ops.conv3x3(xin, wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part, in_stride=Cin + 16, in_coff=16,
                        out_stride=Cout + 8, out_coff=8)
        ops.conv3x3(nhwc(x.detach(), dt), wf, y2, N, H, W, Cin, Cout, ops.TAPS3_FWD, b.to(dev()), None, act=ACT_RELU)
        if Cout % 64 == 0:
            ops.conv3x3(nhwc(dy, dt), wd, dx, N, H, W, Cout, Cin, ops.TAPS3_DGRAD)
        torch.cuda.synchronize()
    finally:
        ops.conv3x3_set_kernel_form(-1)
    assert rel_err(from_nhwc(y[..., 8:]), ref.detach()) < tol(dt)
    assert float(y[..., :8].abs().max()) == 0.0
    p = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()
    s1, s2 = ref.detach().double().sum((0, 2, 3)), (ref.detach().double() ** 2).sum((0, 2, 3))
    assert (p[0] - s1).abs().max() < 1e-3 * max(1.0, s1.abs().max().item())
    assert (p[1] - s2).abs().max() < 1e-3 * s2.abs().max().item()
    assert rel_err(from_nhwc(y2), F.relu(ref.detach() + b.view(1, -1, 1, 1))) < tol(dt)
    if Cout % 64 == 0:
        assert rel_err(from_nhwc(dx), x.grad) < tol(dt)


def test_dma_kernels_repeatable_under_memory_noise():
    """the LDS-DMA kernels read LDS that a DMA filled: a read that overtook its DMA would show up as rare wrong tiles.
    The same launches repeated (every other one racing a large copy on a second stream) must be bit-identical
    (tools/race_screen.py is the long form: 5.5k launches, profiles/r02_race_screen.txt)."""
    from semantic_segmentation_amd import ops
    dt = torch.float16
    side = torch.cuda.Stream()
    src = torch.randn(16 * 1024 * 1024, device=dev())
    dst = torch.empty_like(src)
    for (N, H, W, Cin, Cout) in [(8, 64, 64, 128, 128), (3, 37, 41, 64, 72)]:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev())
        dy = torch.randn(N, H, W, Cout, generator=g).to(dt).to(dev())
        wf = (0.05 * torch.randn(9, Cout, Cin, generator=g)).to(dt).to(dev())
        mt = ops.conv3x3_mtiles(N, H, W, Cout)
        ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), dtype=torch.float32, device=dev())
        try:
            for form in (8, 4):
                ops.conv3x3_set_kernel_form(form)
                ref = None
                for it in range(30):
                    y = torch.empty(N, H, W, Cout, dtype=dt, device=dev())
                    part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev())
                    if it % 2:
                        with torch.cuda.stream(side):
                            dst.copy_(src)
                    ops.conv3x3(x, wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part)
                    if ref is None:
                        ref = (y, part)
                    else:
                        assert torch.equal(y, ref[0]) and torch.equal(part, ref[1]), (form, it)
        finally:
            ops.conv3x3_set_kernel_form(-1)
        refg = None
        for it in range(30):
            gout = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev())
            if it % 2:
                with torch.cuda.stream(side):
                    dst.copy_(src)
            ops.conv3x3_wgrad_det(x, dy, ws, gout, N, H, W, Cin, Cout, 1.0)
            if refg is None:
                refg = gout
            else:
                assert torch.equal(gout, refg), it
        torch.cuda.synchronize()


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,w,Cin,Cout,pad", [(2, 8, 16, 128, 64, (0, 0)), (4, 16, 8, 256, 192, (2, 2)), (8, 32, 32, 128, 64, (1, 0)),
                                                 (1, 16, 16, 384, 128, (0, 0))])
def test_upconv2x2_dma_gemm_path(dtn, dt, N, h, w, Cin, Cout, pad):
    """shapes the LDS-DMA pointwise GEMM (csrc/pwgemm.hip) takes -- power-of-two maps, N*h*w % 256 == 0, Cin % 128 == 0,
    Cout % 64 == 0: several pixel / column tiles, column tiles spanning one, two or four sub-pixel classes, a channel
    slice as input, F.pad offsets, the skip half of the concat buffer untouched"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(16)
    H2, W2 = 2 * h + pad[0], 2 * w + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, Cin, h, w, dt=dt)
    wt = rnd(g, Cin, Cout, 2, 2, dt=dt, scale=0.05)
    b = rnd(g, Cout)
    ref = F.pad(F.conv_transpose2d(x, wt, b, stride=2), [pl, pad[1] - pl, pt, pad[0] - pt])
    wf = torch.empty(4, Cout, Cin, dtype=dt, device=dev())
    ops.pack_weight(wt.to(dev()), wf, None, True)
    xin = torch.full((N, h, w, Cin + 24), 7.0, dtype=dt, device=dev())          # poison around the slice
    xin[..., 8:8 + Cin] = nhwc(x, dt)
    cat = torch.zeros(N, H2, W2, 2 * Cout, dtype=dt, device=dev())
    for _ in range(2):
        ops.upconv2x2_fwd(xin, wf, b.to(dev()), cat, N, 1, h, w, Cin, Cout, 1, H2, W2, in_stride=Cin + 24, in_coff=8,
                          out_stride=2 * Cout, out_coff=Cout, ooy=pt, oox=pl)
    torch.cuda.synchronize()
    got = from_nhwc(cat[..., Cout:])
    inner = torch.zeros_like(ref, dtype=torch.bool)
    inner[:, :, pt:pt + 2 * h, pl:pl + 2 * w] = True
    assert rel_err(got[inner], ref[inner]) < tol(dt)
    assert float(got[~inner].abs().max() if (~inner).any() else 0.0) == 0.0
    assert float(cat[..., :Cout].abs().max()) == 0.0


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,w,Cin,Cout,pad", [(2, 8, 16, 128, 64, (0, 0)), (4, 16, 8, 256, 192, (2, 2)), (8, 32, 32, 128, 64, (1, 0)),
                                                 (1, 16, 16, 384, 128, (0, 0)), (2, 5, 6, 128, 64, (1, 1))])
def test_upconv2x2_dgrad(dtn, dt, N, h, w, Cin, Cout, pad):
    """data gradient of ConvTranspose2d(k2, s2) out of the up half of a concat-gradient buffer: the LDS-DMA GEMM over
    (sub-pixel class, co) on the shapes it covers (a partial last column tile at Cin = 128 / 384), the generic engine on
    the rest (last case) -- both through ops.upconv2x2_dgrad"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(17)
    H2, W2 = 2 * h + pad[0], 2 * w + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, Cin, h, w, dt=dt).requires_grad_(True)
    wt = rnd(g, Cin, Cout, 2, 2, dt=dt, scale=0.05)
    dyf = rnd(g, N, Cout, H2, W2, dt=dt)
    out = F.pad(F.conv_transpose2d(x, wt, None, stride=2), [pl, pad[1] - pl, pt, pad[0] - pt])
    out.backward(dyf)
    wf = torch.empty(4, Cout, Cin, dtype=dt, device=dev())
    wd = torch.empty(4, Cin, Cout, dtype=dt, device=dev())
    ops.pack_weight(wt.to(dev()), wf, wd, True)
    dcat = torch.full((N, H2, W2, 2 * Cout), 3.0, dtype=dt, device=dev())          # the skip half must not be read
    dcat[..., Cout:] = nhwc(dyf, dt)
    taps = [(py + pt, px + pl) for py in range(2) for px in range(2)]
    geom = ops.make_geom(N, H2, W2, Cout, h, w, Cin, h, w, taps, isy=2, isx=2, in_stride=2 * Cout, in_coff=Cout)
    dz = torch.full((N, h, w, Cin), float("nan"), dtype=dt, device=dev())
    ops.upconv2x2_dgrad(geom, dcat, wd, dz, N, h, w, Cin, Cout, H2, W2, 2 * Cout, Cout, pt, pl)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dz), x.grad) < tol(dt)


@pytest.mark.parametrize("dtn,dt", DTS)
@pytest.mark.parametrize("N,h,w,Cin,Cout,pad,pair", [(2, 16, 16, 128, 64, (0, 0), False), (4, 32, 64, 256, 128, (0, 0), True),
                                                      (3, 64, 32, 128, 64, (2, 2), False), (8, 16, 16, 1024, 512, (0, 0), False),
                                                      (1, 8, 16, 256, 192, (1, 1), False), (2, 5, 6, 128, 64, (0, 0), False)])
def test_upconv2x2_wgrad(dtn, dt, N, h, w, Cin, Cout, pad, pair):
    """weight gradient of ConvTranspose2d(k2, s2) (unet_parts.py:51,57) out of the up half of a concat-gradient buffer: the
    K = pixels LDS-DMA GEMM (csrc/upwgrad.hip: several K parts, several (ci, co) tile pairs, padded output, the layer input as
    the hi plane of a pair buffer) on the shapes it covers, the generic engine on the rest (Cout = 192 and the odd map) -- both
    through ops.upconv2x2_wgrad_det, against torch fp32 autograd; bit-reproducible."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(19)
    H2, W2 = 2 * h + pad[0], 2 * w + pad[1]
    pt, pl = pad[0] // 2, pad[1] // 2
    x = rnd(g, N, Cin, h, w, dt=dt)
    wt = rnd(g, Cin, Cout, 2, 2, dt=dt, scale=0.05).requires_grad_(True)
    dyf = rnd(g, N, Cout, H2, W2, dt=dt)
    out = F.pad(F.conv_transpose2d(x, wt, None, stride=2), [pl, pad[1] - pl, pt, pad[0] - pt])
    out.backward(dyf)
    dcat = torch.full((N, H2, W2, 2 * Cout), 3.0, dtype=dt, device=dev())          # the skip half must not be read
    dcat[..., Cout:] = nhwc(dyf, dt)
    xs = 2 * Cin if pair else Cin
    xin = torch.full((N, h, w, xs), 5.0, dtype=dt, device=dev())                    # pair: the lo plane must not be read
    xin[..., :Cin] = nhwc(x, dt)
    taps = [(py + pt, px + pl) for py in range(2) for px in range(2)]
    geom = ops.make_geom(N, H2, W2, Cout, h, w, Cin, h, w, taps, isy=2, isx=2, in_stride=2 * Cout, in_coff=Cout, out_stride=xs)
    need = max(ops.conv_wgrad_ws_floats(geom), ops.upconv2x2_wgrad_ws_floats(N, h, w, Cin, Cout), 1)
    fast = ops.upconv2x2_wgrad_ws_floats(N, h, w, Cin, Cout) > 0
    assert fast == (Cout % 64 == 0 and h & (h - 1) == 0 and w & (w - 1) == 0 and w >= 16 and Cin % 128 == 0)
    res = []
    for rep in range(2):
        ws = torch.full((need,), float("nan"), dtype=torch.float32, device=dev())
        dw = torch.full((Cin, Cout, 2, 2), float("nan"), dtype=torch.float32, device=dev())
        ops.upconv2x2_wgrad_det(geom, xin, dcat, ws, dw, N, h, w, Cin, Cout, H2, W2, xs, 2 * Cout, Cout, pt, pl, 0.5)
        torch.cuda.synchronize()
        res.append(dw.clone())
    assert torch.equal(res[0], res[1])
    assert rel_err(res[0].cpu() * 2, wt.grad) < 2e-3 * (1 if dt == torch.float16 else 4), rel_err(res[0].cpu() * 2, wt.grad)


@pytest.mark.parametrize("dtn,dt", DTS)
def test_pack_weight_multi_matches_single(dtn, dt):
    """gs_pack_weight_multi (all stale conv weights of a network in one launch) == gs_pack_weight per tensor, bit for bit:
    3x3 convs and k2/s2 transposed convs of mixed sizes, with and without the data-gradient pack"""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(21)
    shapes = [((64, 64, 3, 3), False), ((128, 64, 3, 3), False), ((72, 40, 3, 3), False), ((256, 128, 2, 2), True),
              ((1024, 512, 2, 2), True), ((512, 1024, 3, 3), False)]
    items, singles = [], []
    for k, (shp, tr) in enumerate(shapes):
        w = torch.randn(*shp, generator=g).to(dev())
        cin, cout = (shp[0], shp[1]) if tr else (shp[1], shp[0])
        taps = shp[2] * shp[3]
        need_d = k % 3 != 1
        wf = torch.full((taps, cout, cin), 7.0, dtype=dt, device=dev())
        wd = torch.full((taps, cin, cout), 7.0, dtype=dt, device=dev()) if need_d else None
        wf1 = torch.empty_like(wf)
        wd1 = torch.empty_like(wd) if need_d else None
        ops.pack_weight(w, wf1, wd1, tr)
        items.append((w, wf, wd, tr))
        singles.append((wf1, wd1))
    ops.pack_weight_multi(items)
    torch.cuda.synchronize()
    for (w, wf, wd, tr), (wf1, wd1) in zip(items, singles):
        assert torch.equal(wf, wf1)
        if wd is not None:
            assert torch.equal(wd, wd1)


# ------------------------------------------------------------------------------------------------ "q" stages (FP8 correction segment)
@pytest.mark.parametrize("N,H,W,cin,cout,form", [(2, 32, 32, 64, 64, -1), (1, 24, 40, 128, 64, 8), (2, 16, 32, 64, 128, 4), (1, 16, 64, 256, 128, -1)])
def test_conv3x3_q8_matches_its_arithmetic(N, H, W, cin, cout, form):
    """gs_conv3x3_q8 (x_hi.w_hi on the 16-bit MFMA + ONE block-scaled e4m3 segment for x_lo.w_hi + x_hi.w_lo) against the SAME
    arithmetic in fp32 torch: q planes written by gs_bn_act_apply_split_q8, q pack + per-cout exponents by gs_pack_weight_q8,
    e4m3 values emulated with torch.float8_e4m3fn (saturating).  The result must also be >= 10x closer to the exact fp32
    convolution than the plain 16-bit product; BatchNorm partial rows sum to the column sums.  8- and 4-wave forms."""
    import torch.nn.functional as F
    from semantic_segmentation_amd import ops
    dev = torch.device("cuda:0")
    E4 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(cin + cout + H)
    v = torch.randn(N, H, W, cin, generator=g).abs() * 1.5
    v[0, 0, 0, :4] = torch.tensor([3000.0, 1e-4, 0.0, 500.0])                   # saturating / tiny / zero activations
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * (1.0 / (cin * 9) ** 0.5)
    wt[3] *= 40.0                                                                # rows of very different magnitude: per-cout exponents
    wt[5] *= 1e-3
    yh = v.half().to(dev).contiguous()
    yl = (v - v.half().float()).half().to(dev).contiguous()
    one, zero = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    xbuf = torch.zeros(N, H, W, 2 * cin, dtype=torch.float16, device=dev)
    ops.bn_act_apply_split_q8(yh, yl, one, zero, 0, xbuf, xbuf[..., cin:], True, 2 * cin, 0)
    pack = torch.empty(9, cout, 2 * cin, dtype=torch.float16, device=dev)
    wexp = torch.empty(cout, dtype=torch.int32, device=dev)
    ops.pack_weight_q8([(wt.to(dev).contiguous(), pack, wexp)])
    y_hi = torch.empty(N, H, W, cout, dtype=torch.float16, device=dev)
    y_lo = torch.empty_like(y_hi)
    ops.conv3x3_set_kernel_form(form)
    try:
        nt = ops.conv3x3_stat_rows(N, H, W, 2 * cin, cout, pair="q")
        part = torch.zeros(ops.bn_partials_numel(nt, cout), dtype=torch.float32, device=dev)
        ops.conv3x3_q8(xbuf, pack, wexp, y_hi, y_lo, N, H, W, cin, cout, 2 * cin, 0, part)
        torch.cuda.synchronize()
    finally:
        ops.conv3x3_set_kernel_form(-1)
    got = (y_hi.float() + y_lo.float()).cpu().permute(0, 3, 1, 2)

    def q8(t, k):
        s = 2.0 ** k
        return (t * s).clamp(-448, 448).to(E4).float() / s
    vv = (yh.float() + yl.float()).cpu()
    hi2 = vv.half().float()
    xh, xl = hi2.permute(0, 3, 1, 2), (vv - hi2).permute(0, 3, 1, 2)
    wh = wt.half().float()
    wl = wt - wh
    e = wexp.cpu().view(-1, 1, 1, 1).double()
    amax = wt.abs().flatten(1).amax(1)
    assert torch.all((amax * 2.0 ** wexp.cpu().float() >= 64) & (amax * 2.0 ** wexp.cpu().float() < 128)), "per-cout exponents"
    wh8 = ((wh.double() * 2 ** e).clamp(-448, 448).float().to(E4).double() / 2 ** e).float()
    wl8 = ((wl.double() * 2 ** (e + 11)).clamp(-448, 448).float().to(E4).double() / 2 ** (e + 11)).float()
    ref = F.conv2d(xh, wh, padding=1) + F.conv2d(q8(xl, 9), wh8, padding=1) + F.conv2d(q8(xh, -2), wl8, padding=1)
    exact = F.conv2d(vv.permute(0, 3, 1, 2), wt, padding=1)
    plain = F.conv2d(xh, wh, padding=1)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-6 * scale + 2e-6, float((got - ref).abs().max())
    # away from the saturated pixel the correction segment recovers the fp32 product to ~2^-15
    m = torch.ones_like(exact, dtype=torch.bool)
    m[0, :, :2, :2] = False
    assert float((got - exact)[m].abs().max()) * 10 < float((plain - exact)[m].abs().max())
    s1 = part[:nt * 2 * cout].view(nt, 2, cout)[:, 0].sum(0).cpu()
    assert float((s1 - got.sum((0, 2, 3))).abs().max()) < 1e-5 * float(got.abs().sum((0, 2, 3)).max())


def test_q8_from_hi_and_stem_q8_planes():
    """the two other producers of q planes: gs_q8_from_hi (hi8 from a stored hi plane, lo8 = 0) and gs_stem_fwd_bn_pair_q8"""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    dev = torch.device("cuda:0")
    E4 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(3)
    N, H, W, C = 2, 6, 10, 64
    buf = torch.zeros(N, H, W, 4 * C, dtype=torch.float16, device=dev)              # [skip_h up_h | q plane]
    up = torch.randn(N, H, W, C, generator=g).half()
    buf[..., C:2 * C] = up.to(dev)
    ops.q8_from_hi(buf, buf[..., 2 * C:], N * H * W, C, 4 * C, C)
    q = buf[..., 2 * C:].contiguous().view(torch.uint8).view(N, H, W, 4 * C)           # bytes of the q plane: chunks of 64 per 32 channels
    chunks = q.view(N, H, W, 4, 64).cpu()
    assert int(chunks[..., :2, :].abs().sum()) == 0                                   # the skip half was not touched
    hi8 = chunks[..., 2:, 32:].reshape(N, H, W, C).view(E4).float()
    lo8 = chunks[..., 2:, :32].reshape(N, H, W, C)
    assert int(lo8.abs().sum()) == 0
    assert torch.equal(hi8, (up.float() * 0.25).clamp(-448, 448).to(E4).float())
    # stem: [hi | q plane] against the 16-bit pair form
    x = torch.randn(2, 1, 32, 40, generator=g).to(dev)
    w = (torch.randn(64, 1, 3, 3, generator=g) * 0.3).to(dev)
    sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    zq = torch.empty(2, 32, 40, 128, dtype=torch.float16, device=dev)
    zp = torch.empty_like(zq)
    ops.stem_fwd_bn_pair_q8(x, w, sc, sh, ACT_RELU, zq)
    ops.stem_fwd_bn_pair(x, w, sc, sh, ACT_RELU, zp, write_lo=True)
    assert torch.equal(zq[..., :64], zp[..., :64])
    hi, lo = zp[..., :64].float().cpu(), zp[..., 64:].float().cpu()
    ch = zq[..., 64:].contiguous().view(torch.uint8).view(2, 32, 40, 2, 64).cpu()
    assert torch.equal(ch[..., 32:].reshape(2, 32, 40, 64).view(E4).float(), (hi * 0.25).clamp(-448, 448).to(E4).float())
    got_lo = ch[..., :32].reshape(2, 32, 40, 64).view(E4).float() / 512.0
    assert float((got_lo - lo).abs().max()) <= float(lo.abs().max()) / 16 + 2.0 ** -18          # e4m3 of (value - hi) * 2^9


@pytest.mark.parametrize("N,H,W,C,coff", [(2, 16, 24, 64, 0), (1, 17, 31, 128, 0), (3, 8, 8, 8, 0)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_maxpool2x2_pair_matches_its_definition(N, H, W, C, coff, dtype):
    """gs_maxpool2x2_fwd_pair (nn.MaxPool2d(2) of `Down`, unet_parts.py:34, on hi/lo pairs): the pooled pair is the maximum of the
    window's pair VALUES hi + lo, split again -- bit-exact against that definition in torch; odd sizes drop the last row / column."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(5)
    v = torch.randn(N, H, W, C, generator=g)
    hi = v.to(dtype)
    lo = (v - hi.float()).to(dtype)
    buf = torch.empty(N, H, W, 2 * C + 16, dtype=dtype)
    buf[..., :C], buf[..., C + 8:2 * C + 8] = hi, lo                       # two planes of one buffer, pixel stride 2C + 16
    buf = buf.cuda()
    zp = torch.zeros(N, H // 2, W // 2, 2 * C, dtype=dtype, device="cuda")
    ops.maxpool2x2_fwd_pair(buf, buf[..., C + 8:], 2 * C + 16, zp, zp[..., C:], 2 * C, N, H, W, C)
    val = (hi.float() + lo.float())[:, :H // 2 * 2, :W // 2 * 2]
    m = val.reshape(N, H // 2, 2, W // 2, 2, C).amax(dim=(2, 4))
    mh = m.to(dtype)
    ml = (m - mh.float()).to(dtype)
    assert torch.equal(zp[..., :C].cpu(), mh) and torch.equal(zp[..., C:].cpu(), ml)


@pytest.mark.parametrize("N,Cin,H,W", [(6, 3, 20, 28), (2, 1, 33, 17), (16, 3, 64, 64)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_smallcin_split_cout32_thread_per_pixel_form(N, Cin, H, W, dtype):
    """gs_conv_smallcin_fwd_split at Cout = 32 (the first conv of UNet3D, unet3d.py:28, as a 2-D conv over three depth slices) runs
    a one-thread-per-pixel kernel with LDS-assembled complete-line stores: same tap order as the generic kernel, so the pair is
    BIT-IDENTICAL to it (GSSEG_SC32_OFF=1 selects the generic one); the BatchNorm partial rows agree in their totals; and the
    pair value matches fp32 F.conv2d."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (0.3 * torch.randn(32, Cin, 3, 3, generator=g)).cuda()
    nt = ops.conv_smallcin_mtiles(N, H, W)
    out = {}
    for off in (True, False):
        if off:
            os.environ["GSSEG_SC32_OFF"] = "1"
        else:
            os.environ.pop("GSSEG_SC32_OFF", None)
        try:
            yh = torch.zeros(N, H, W, 32, dtype=dtype, device="cuda")
            yl = torch.zeros_like(yh)
            part = torch.zeros(ops.bn_partials_numel(nt, 32), dtype=torch.float32, device="cuda")
            ops.conv_smallcin_fwd_split(x, w, yh, yl, part, 3, 1)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("GSSEG_SC32_OFF", None)
        out[off] = (yh, yl, part[:nt * 64].view(nt, 2, 32).double().sum(0))
    assert torch.equal(out[True][0].view(torch.int16), out[False][0].view(torch.int16))
    assert torch.equal(out[True][1].view(torch.int16), out[False][1].view(torch.int16))
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1)
    val = out[False][0].float() + out[False][1].float()
    assert float((val - ref).abs().max()) < (2e-5 if dtype == torch.float16 else 2e-4) * max(1.0, float(ref.abs().max()))
    s_ref = torch.stack([ref.double().sum((0, 1, 2)), (ref.double() ** 2).sum((0, 1, 2))])
    for k in (True, False):
        assert float((out[k][2] - s_ref).abs().max()) < 1e-4 * float(s_ref.abs().max())


@pytest.mark.parametrize("N,h,w,Cin", [(2, 16, 24, 128), (1, 9, 70, 128), (3, 8, 8, 256)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_upconv8_image_wgrad_matches_autograd(N, h, w, Cin, dtype):
    """gs_upconv8_image_wgrad: weight gradient of the merged 8x8 / stride 2 / pad 3 transposed conv to ONE image channel (the
    outermost up path of the JSRT generator, networks.py:486-511,588-593) against torch autograd of F.conv_transpose2d on the same
    16-bit-rounded operands, in the [4 classes][16 taps][1][Cin] layout gs_upconv_split_wgrad reads; deterministic (two runs equal)."""
    from semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, h, w, Cin, generator=g).to(dtype)
    du = torch.zeros(N, 2 * h, 2 * w, 8, dtype=dtype)
    du[..., 0] = torch.randn(N, 2 * h, 2 * w, generator=g).to(dtype)
    du[..., 1:] = 7.0                                                   # the padding channels must not be read
    assert ops.upconv8_image_wgrad_ok(Cin, 1) and not ops.upconv8_image_wgrad_ok(Cin, 3) and not ops.upconv8_image_wgrad_ok(64, 1)
    outs = []
    for _ in range(2):
        dwm = torch.full((4, 16, 1, Cin), float("nan"), dtype=torch.float32, device="cuda")
        ops.upconv8_image_wgrad(x.cuda(), du.cuda(), dwm, N, h, w, Cin)
        outs.append(dwm.cpu())
    assert torch.equal(outs[0], outs[1])
    wm = torch.zeros(Cin, 1, 8, 8, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wm, stride=2, padding=3)
    (y * du[..., 0].double().unsqueeze(1)).sum().backward()
    ref = torch.empty(4, 16, 1, Cin, dtype=torch.float64)
    for ky in range(8):
        for kx in range(8):
            cls, t16 = 2 * (1 - ky % 2) + (1 - kx % 2), 4 * (ky // 2) + kx // 2
            ref[cls, t16, 0] = wm.grad[:, 0, ky, kx]
    err = float((outs[0].double() - ref).abs().max())
    assert err < 2e-4 * float(ref.abs().max()) + 1e-4, err


@pytest.mark.parametrize("NB,D,H,W,C", [(1, 4, 6, 8, 64), (2, 2, 4, 10, 32), (1, 8, 8, 8, 128)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_bn_act_apply_split_pool3d_equals_apply_then_pool(NB, D, H, W, C, dtype):
    """gs_bn_act_apply_split_pool3d (BatchNorm3d + ReLU + MaxPool3d(2) of a UNet3D analysis block, unet3d.py:29-36, on pairs) is
    BIT-IDENTICAL to gs_bn_act_apply_split followed by gs_maxpool3d_fwd_pair: the z pair inside a wider concat buffer and the pooled pair."""
    from semantic_segmentation_amd import ops
    from semantic_segmentation_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(2)
    v = torch.randn(NB * D, H, W, C, generator=g)
    yh = v.to(dtype)
    yl = (v - yh.float()).to(dtype)
    yh, yl = yh.cuda(), yl.cuda()
    sc = (torch.rand(C, generator=g) + 0.5).cuda()
    sh = (torch.randn(C, generator=g) * 0.2).cuda()
    cu = 16                                                # the residual sits behind `cu` up channels: [up_h res_h | res_l]
    res = []
    for fused in (False, True):
        cat = torch.zeros(NB * D, H, W, 2 * (cu + C), dtype=dtype, device="cuda")
        pooled = torch.zeros(NB * D // 2, H // 2, W // 2, 2 * C, dtype=dtype, device="cuda")
        if fused:
            ops.bn_act_apply_split_pool3d(yh, yl, sc, sh, ACT_RELU, cat, cat[..., C:], 2 * (cu + C), cu, pooled, pooled[..., C:], 2 * C,
                                          NB, D, H, W)
        else:
            ops.bn_act_apply_split(yh, yl, sc, sh, ACT_RELU, cat, cat[..., C:], 2 * (cu + C), cu)
            ops.maxpool3d_fwd_pair(cat[..., cu:], cat[..., cu + C:], 2 * (cu + C), pooled, pooled[..., C:], 2 * C, NB, D, H, W, C)
        torch.cuda.synchronize()
        res.append((cat.cpu(), pooled.cpu()))
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
    assert torch.equal(res[0][1].view(torch.int16), res[1][1].view(torch.int16))
    assert float(res[1][1].float().abs().max()) > 0
