#!/usr/bin/env python3
"""Race screen for the LDS-DMA kernels (conv3x3 forms 4 / 8 / 2, wgrad3x3): the same launch repeated many times at several
sizes must give bit-identical outputs every time (an LDS read that overtakes its DMA shows up as rare wrong tiles), also
while a second stream keeps the memory system busy.  Run on the GPU box:  python tools/race_screen.py [--iters 200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import ops  # noqa: E402

SHAPES = [(32, 256, 256, 64, 64), (32, 128, 128, 128, 128), (32, 64, 64, 256, 256), (32, 32, 32, 1024, 512), (3, 37, 41, 64, 72),
          (2, 256, 256, 128, 64), (7, 50, 70, 192, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16
    side = torch.cuda.Stream()
    noise_src = torch.randn(64 * 1024 * 1024, device=dev)
    noise_dst = torch.empty_like(noise_src)
    bad = 0
    for (N, H, W, Cin, Cout) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
        dy = torch.randn(N, H, W, Cout, generator=g).to(dt).to(dev)
        wf = (0.05 * torch.randn(9, Cout, Cin, generator=g)).to(dt).to(dev)
        mt = ops.conv3x3_mtiles(N, H, W, Cout)
        for form in (8, 4):
            ops.conv3x3_set_kernel_form(form)
            ref_y = ref_p = None
            mism = 0
            for it in range(a.iters):
                y = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
                part = torch.zeros(ops.bn_partials_numel(mt, Cout), dtype=torch.float32, device=dev)
                if it % 2:                               # every other launch races a big copy on a second stream
                    with torch.cuda.stream(side):
                        noise_dst.copy_(noise_src)
                ops.conv3x3(x, wf, y, N, H, W, Cin, Cout, ops.TAPS3_FWD, None, part)
                if ref_y is None:
                    ref_y, ref_p = y, part
                elif not (torch.equal(y, ref_y) and torch.equal(part, ref_p)):
                    mism += 1
            torch.cuda.synchronize()
            bad += mism
            print(f"conv3x3 form {form} {(N, H, W, Cin, Cout)}: {mism} of {a.iters - 1} repeats differ", flush=True)
        ops.conv3x3_set_kernel_form(-1)
        ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), dtype=torch.float32, device=dev)
        ref_g = None
        mism = 0
        for it in range(a.iters):
            gout = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev)
            if it % 2:
                with torch.cuda.stream(side):
                    noise_dst.copy_(noise_src)
            ops.conv3x3_wgrad_det(x, dy, ws, gout, N, H, W, Cin, Cout, 1.0)
            if ref_g is None:
                ref_g = gout
            elif not torch.equal(gout, ref_g):
                mism += 1
        torch.cuda.synchronize()
        bad += mism
        print(f"wgrad3x3 {(N, H, W, Cin, Cout)}: {mism} of {a.iters - 1} repeats differ", flush=True)
    print("TOTAL MISMATCHES", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
