#!/usr/bin/env python3
"""Per-layer micro-benchmark of the U-Net convolutions (BASELINE config 2 shapes, batch 32) on one MI355X:
forward (halo kernel vs generic gather engine), data gradient and weight gradient, in TFLOP/s.
Run on the GPU box:  python tools/bench_layers.py [--batch 32] [--dtype f16]"""
import argparse
import os
import sys

import os
os.environ.setdefault("GSSEG_C3_DEBUG", "1")   # honour the ablation bits of gs_conv3x3 in this tool
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import ops  # noqa: E402

LAYERS = [  # name, H, Cin, Cout
    ("inc.3", 256, 64, 64), ("d1.0", 128, 64, 128), ("d1.3", 128, 128, 128), ("d2.0", 64, 128, 256),
    ("d2.3", 64, 256, 256), ("d3.0", 32, 256, 512), ("d3.3", 32, 512, 512), ("d4.0", 16, 512, 1024),
    ("d4.3", 16, 1024, 1024), ("u1.0", 32, 1024, 512), ("u2.0", 64, 512, 256), ("u3.0", 128, 256, 128),
    ("u4.0", 256, 128, 64),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    N = a.batch
    print(f"{'layer':8s} {'GF':>8s} | {'halo fwd':>9s} {'gen fwd':>9s} {'halo dgr':>9s} {'gen dgr':>9s} {'gen wgr':>9s} {'halo wgr':>9s}  (TFLOP/s)")
    tot = {"halo": 0.0, "gen": 0.0, "wgrad": 0.0, "hw": 0.0}
    for name, H, Cin, Cout in LAYERS:
        if a.only and a.only not in name:
            continue
        x = torch.randn(N, H, H, Cin, device=dev).to(dt)
        dy = torch.randn(N, H, H, Cout, device=dev).to(dt)
        wf = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
        wd = (0.05 * torch.randn(9, Cin, Cout, device=dev)).to(dt)
        y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
        dx = torch.empty(N, H, H, Cin, device=dev, dtype=dt)
        dw = torch.zeros(9, Cout, Cin, device=dev)
        part = torch.empty(max(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout),
                               ops.bn_partials_numel((N * H * H + 127) // 128, Cout)), device=dev)
        g = ops.geom_conv(N, H, H, Cin, Cout, 3, 1, 1)
        gd = ops.geom_conv_dgrad_s1(N, H, H, Cin, Cout, 3, 1)
        fl = 2.0 * N * H * H * Cout * Cin * 9
        t_hf = timeit(lambda: ops.conv3x3(x, wf, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part))
        t_gf = timeit(lambda: ops.conv_igemm(g, x, wf, y, None, part))
        t_hd = timeit(lambda: ops.conv3x3(dy, wd, dx, N, H, H, Cout, Cin, ops.TAPS3_DGRAD))
        t_gd = timeit(lambda: ops.conv_igemm(gd, dy, wd, dx))
        t_w = timeit(lambda: ops.conv_wgrad(g, x, dy, dw))
        t_hw = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, N, H, H, Cin, Cout))
        tot["halo"] += t_hf + t_hd; tot["gen"] += t_gf + t_gd; tot["wgrad"] += t_w; tot["hw"] += t_hw
        print(f"{name:8s} {fl / 1e9:8.1f} | {fl / t_hf / 1e12:9.1f} {fl / t_gf / 1e12:9.1f} {fl / t_hd / 1e12:9.1f} "
              f"{fl / t_gd / 1e12:9.1f} {fl / t_w / 1e12:9.1f} {fl / t_hw / 1e12:9.1f}")
    print("sum ms: halo fwd+dgrad %.3f  generic fwd+dgrad %.3f  generic wgrad %.3f  halo wgrad %.3f"
          % (tot["halo"] * 1e3, tot["gen"] * 1e3, tot["wgrad"] * 1e3, tot["hw"] * 1e3))


if __name__ == "__main__":
    main()
