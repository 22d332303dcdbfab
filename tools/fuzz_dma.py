#!/usr/bin/env python3
"""Random-shape fuzz of the LDS-DMA kernels against fp32 torch on the same 16-bit-rounded inputs (run on the GPU box):
conv3x3 in every kernel form (forward with BatchNorm partials into a strided slice, bias + ReLU epilogue, data gradient),
the deterministic 3x3 weight gradient, and the up-conv forward / data gradient GEMM.
    python tools/fuzz_dma.py [--cases 60] [--seed 0]"""
import argparse
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import ops  # noqa: E402
from semantic_segmentation_amd._lib import ACT_RELU  # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def nhwc(t, dt):
    return t.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def back(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    bad = 0
    for case in range(a.cases):
        dt = rng.choice([torch.float16, torch.bfloat16])
        tol = 3e-3 if dt == torch.float16 else 1.5e-2
        form = rng.choice([-1, 4, 8])
        N, H, W = rng.randint(1, 6), rng.randint(1, 72), rng.randint(24, 80)
        Cin, Cout = 64 * rng.randint(1, 5), rng.choice([64, 128, 256, 8 * rng.randint(1, 30)])
        x = torch.randn(N, Cin, H, W, generator=g).to(dt).float().requires_grad_(True)
        w = (0.05 * torch.randn(Cout, Cin, 3, 3, generator=g)).to(dt).float().requires_grad_(True)
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, None, padding=1)
        dy = torch.randn(N, Cout, H, W, generator=g).to(dt).float()
        ref.backward(dy)
        wf = torch.empty(9, Cout, Cin, dtype=dt, device=DEV)
        wd = torch.empty(9, Cin, Cout, dtype=dt, device=DEV)
        ops.pack_weight(w.detach().to(DEV), wf, wd, False)
        pad = 8 * rng.randint(0, 2)
        ipad = 8 * rng.randint(0, 2)
        y = torch.zeros(N, H, W, Cout + pad, dtype=dt, device=DEV)
        xin = torch.full((N, H, W, Cin + ipad), 5.0, dtype=dt, device=DEV)
        xin[..., ipad:] = nhwc(x.detach(), dt)
        mt = ops.conv3x3_mtiles(N, H, W, Cout)
        part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=DEV)
        y2 = torch.empty(N, H, W, Cout, dtype=dt, device=DEV)
        dx = torch.full((N, H, W, Cin), float("nan"), dtype=dt, device=DEV)
        ops.conv3x3_set_kernel_form(form)
        try:
            ops.conv3x3(xin, wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part, in_stride=Cin + ipad, in_coff=ipad,
                        out_stride=Cout + pad, out_coff=pad)
            ops.conv3x3(nhwc(x.detach(), dt), wf, y2, N, H, W, Cin, Cout, ops.TAPS3_FWD, b.to(DEV), None, act=ACT_RELU)
            do_dgrad = Cout % 64 == 0
            if do_dgrad:
                ops.conv3x3(nhwc(dy, dt), wd, dx, N, H, W, Cout, Cin, ops.TAPS3_DGRAD)
        finally:
            ops.conv3x3_set_kernel_form(-1)
        ws = torch.full((ops.conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout),), float("nan"), dtype=torch.float32, device=DEV)
        gw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=DEV)
        ops.conv3x3_wgrad_det(xin, nhwc(dy, dt), ws, gw, N, H, W, Cin, Cout, 1.0, in_stride=Cin + ipad, in_coff=ipad)
        torch.cuda.synchronize()
        e = [rel(back(y[..., pad:]), ref.detach()), rel(back(y2), F.relu(ref.detach() + b.view(1, -1, 1, 1))),
             rel(back(dx), x.grad) if do_dgrad else 0.0, rel(gw.cpu(), w.grad)]
        p = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()
        s1, s2 = ref.detach().double().sum((0, 2, 3)), (ref.detach().double() ** 2).sum((0, 2, 3))
        es = max(float((p[0] - s1).abs().max() / max(1.0, s1.abs().max().item())), float((p[1] - s2).abs().max() / s2.abs().max().item()))
        z = float(y[..., :pad].abs().max()) if pad else 0.0
        ok = max(e[:3]) < tol and e[3] < 2e-3 and es < 1e-3 and z == 0.0
        bad += 0 if ok else 1
        print(f"{'OK ' if ok else 'BAD'} conv form {form:2d} {str(dt)[6:]:8s} {(N, H, W, Cin, Cout)} fwd {e[0]:.1e} bias-relu {e[1]:.1e} dgrad {e[2]:.1e} "
              f"wgrad {e[3]:.1e} stats {es:.1e}", flush=True)
        # up-conv GEMM: power-of-two maps
        h, wdt = 1 << rng.randint(2, 5), 1 << rng.randint(2, 5)
        n2 = rng.choice([1, 2, 4, 8])
        while n2 * h * wdt % 256:
            n2 *= 2
        cin, cout = 128 * rng.randint(1, 4), 64 * rng.randint(1, 3)
        py, px = rng.randint(0, 2), rng.randint(0, 2)
        H2, W2 = 2 * h + py, 2 * wdt + px
        pt, pl = py // 2, px // 2
        xu = torch.randn(n2, cin, h, wdt, generator=g).to(dt).float().requires_grad_(True)
        wt = (0.05 * torch.randn(cin, cout, 2, 2, generator=g)).to(dt).float()
        bu = torch.randn(cout, generator=g)
        out = F.pad(F.conv_transpose2d(xu, wt, bu, stride=2), [pl, px - pl, pt, py - pt])
        dyu = torch.randn(n2, cout, H2, W2, generator=g).to(dt).float()
        out.backward(dyu)
        wfu = torch.empty(4, cout, cin, dtype=dt, device=DEV)
        wdu = torch.empty(4, cin, cout, dtype=dt, device=DEV)
        ops.pack_weight(wt.to(DEV), wfu, wdu, True)
        cat = torch.zeros(n2, H2, W2, 2 * cout, dtype=dt, device=DEV)
        ops.upconv2x2_fwd(nhwc(xu.detach(), dt), wfu, bu.to(DEV), cat, n2, 1, h, wdt, cin, cout, 1, H2, W2, out_stride=2 * cout,
                          out_coff=cout, ooy=pt, oox=pl)
        dcat = torch.full((n2, H2, W2, 2 * cout), 3.0, dtype=dt, device=DEV)
        dcat[..., cout:] = nhwc(dyu, dt)
        taps = [(a_ + pt, b_ + pl) for a_ in range(2) for b_ in range(2)]
        geom = ops.make_geom(n2, H2, W2, cout, h, wdt, cin, h, wdt, taps, isy=2, isx=2, in_stride=2 * cout, in_coff=cout)
        dz = torch.full((n2, h, wdt, cin), float("nan"), dtype=dt, device=DEV)
        ops.upconv2x2_dgrad(geom, dcat, wdu, dz, n2, h, wdt, cin, cout, H2, W2, 2 * cout, cout, pt, pl)
        torch.cuda.synchronize()
        inner = torch.zeros_like(out, dtype=torch.bool)
        inner[:, :, pt:pt + 2 * h, pl:pl + 2 * wdt] = True
        got = back(cat[..., cout:])
        eu = [rel(got[inner], out.detach()[inner]), rel(back(dz), xu.grad)]
        oku = max(eu) < tol and float(cat[..., :cout].abs().max()) == 0.0 and (not (~inner).any() or float(got[~inner].abs().max()) == 0.0)
        bad += 0 if oku else 1
        print(f"{'OK ' if oku else 'BAD'} upconv {str(dt)[6:]:8s} {(n2, h, wdt, cin, cout)} pad {(py, px)} fwd {eu[0]:.1e} dgrad {eu[1]:.1e}", flush=True)
    print("FAILURES", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
