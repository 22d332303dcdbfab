#!/usr/bin/env python3
"""Random-shape fuzz of the MFMA kernels against fp32 torch on the same 16-bit-rounded inputs (run on the GPU box):
3x3 halo conv forward (+BN partials, strided output) / data gradient / deterministic weight gradient, the generic
implicit-GEMM conv (k 1..4, stride 1..2) with its weight gradient, and the stand-alone 2x2 pool.
    python tools/fuzz_kernels.py [--cases 40] [--seed 0]"""
import argparse
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import ops  # noqa: E402

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
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    worst = 0.0
    for case in range(a.cases):
        dt = rng.choice([torch.float16, torch.bfloat16])
        tol = 3e-3 if dt == torch.float16 else 1.5e-2
        N, H, W = rng.randint(1, 5), rng.randint(1, 70), rng.randint(1, 70)
        Cin, Cout = 8 * rng.randint(1, 40), 8 * rng.randint(1, 40)
        x = (torch.randn(N, Cin, H, W, generator=g)).to(dt).float().requires_grad_(True)
        w = (0.05 * torch.randn(Cout, Cin, 3, 3, generator=g)).to(dt).float().requires_grad_(True)
        ref = F.conv2d(x, w, None, padding=1)
        dy = torch.randn(N, Cout, H, W, generator=g).to(dt).float()
        ref.backward(dy)
        wf = torch.empty(9, Cout, Cin, dtype=dt, device=DEV)
        wd = torch.empty(9, Cin, Cout, dtype=dt, device=DEV)
        ops.pack_weight(w.detach().to(DEV), wf, wd, False)
        pad = 8 * rng.randint(0, 2)
        y = torch.zeros(N, H, W, Cout + pad, dtype=dt, device=DEV)
        mt = ops.conv3x3_mtiles(N, H, W, Cout)
        part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=DEV)
        xd, dyd = nhwc(x.detach(), dt), nhwc(dy, dt)
        ops.conv3x3(xd, wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part, out_stride=Cout + pad, out_coff=pad)
        dx = torch.empty(N, H, W, Cin, dtype=dt, device=DEV)
        ops.conv3x3(dyd, wd, dx, N, H, W, Cout, Cin, ops.TAPS3_DGRAD)
        ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), dtype=torch.float32, device=DEV)
        dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=DEV)
        ops.conv3x3_wgrad_det(xd, dyd, ws, dw, N, H, W, Cin, Cout, 1.0)
        torch.cuda.synchronize()
        e = [rel(back(y[..., pad:]), ref.detach()), rel(back(dx), x.grad), rel(dw, w.grad)]
        s1 = part[: mt * 2 * Cout].view(mt, 2, Cout).double().sum(0).cpu()[0]
        e.append(float((s1 - ref.detach().double().sum((0, 2, 3))).abs().max() / max(1.0, float(ref.detach().double().sum((0, 2, 3)).abs().max()))))
        if pad:
            assert float(y[..., :pad].abs().max()) == 0.0, "halo conv wrote outside its channel slice"
        # generic engine: random k / stride
        k, s = rng.choice([(1, 1), (2, 2), (3, 1), (3, 2), (4, 2), (4, 1)])
        p = rng.randint(0, k // 2)
        if H + 2 * p >= k and W + 2 * p >= k:
            w2 = (0.05 * torch.randn(Cout, Cin, k, k, generator=g)).to(dt).float().requires_grad_(True)
            x2 = x.detach().clone().requires_grad_(True)
            r2 = F.conv2d(x2, w2, None, stride=s, padding=p)
            dy2 = torch.randn(*r2.shape, generator=g).to(dt).float()
            r2.backward(dy2)
            geom = ops.geom_conv(N, H, W, Cin, Cout, k, s, p)
            wf2 = torch.empty(k * k, Cout, Cin, dtype=dt, device=DEV)
            ops.pack_weight(w2.detach().to(DEV), wf2, None, False)
            y2 = torch.empty(N, r2.shape[2], r2.shape[3], Cout, dtype=dt, device=DEV)
            ops.conv_igemm(geom, xd, wf2, y2, None, None)
            dw2 = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=DEV)
            ws2 = torch.empty(ops.conv_wgrad_ws_floats(geom), dtype=torch.float32, device=DEV)
            ops.conv_wgrad_det(geom, xd, nhwc(dy2, dt), ws2, dw2, Cout, Cin, k * k, 1.0)
            torch.cuda.synchronize()
            e += [rel(back(y2), r2.detach()), rel(dw2, w2.grad)]
        if H >= 2 and W >= 2:
            zp = torch.empty(N, H // 2, W // 2, Cin, dtype=dt, device=DEV)
            ops.maxpool2x2_fwd(xd, zp, N, H, W, Cin)
            torch.cuda.synchronize()
            assert torch.equal(back(zp), F.max_pool2d(x.detach().to(dt).float(), 2)), "maxpool mismatch"
        bad = max(e) >= tol
        worst = max(worst, max(e) / tol)
        print(f"case {case:3d} {str(dt)[6:]:8s} N={N} H={H} W={W} Cin={Cin} Cout={Cout} k={k} s={s} p={p}: " +
              " ".join(f"{v:.1e}" for v in e) + ("   <-- FAIL" if bad else ""), flush=True)
        if bad:
            raise SystemExit(1)
    print(f"fuzz OK: {a.cases} cases, worst error = {worst:.2f} x tolerance")


if __name__ == "__main__":
    main()
