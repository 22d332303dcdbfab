#!/usr/bin/env python3
"""Per-level micro-benchmark of the BatchNorm / activation passes (BASELINE config 2 shapes, batch 32): achieved HBM GB/s
of gs_bn_act_apply (plain / pooled), gs_bn_act_bwd_reduce and gs_bn_act_bwd_apply.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import ops  # noqa: E402
from semantic_segmentation_amd._lib import ACT_RELU  # noqa: E402


def timeit(fn, iters=10):
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
    dev = torch.device("cuda:0")
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dt = torch.float16
    print(f"{'level':>14s} {'MB/tensor':>9s} | {'apply':>8s} {'apply+pool':>10s} {'bwd reduce':>10s} {'bwd apply':>10s} {'bwd red(pool)':>13s} {'bwd app(pool)':>13s}   GB/s (us)")
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for H, C in ((256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)):
        y = torch.randn(N, H, H, C, device=dev).to(dt)
        dz = torch.randn(N, H, H, C, device=dev).to(dt)
        dzp = torch.randn(N, H // 2, H // 2, C, device=dev).to(dt)
        z = torch.empty(N, H, H, C, device=dev, dtype=dt)
        zp = torch.empty(N, H // 2, H // 2, C, device=dev, dtype=dt)
        dy = torch.empty_like(y)
        coef = torch.rand(4, C, device=dev) + 0.5
        c12 = torch.rand(2, C, device=dev) * 0.01
        part = torch.empty(ops.bn_partials_numel(ops.bn_bwd_tiles(N, H, H), C), device=dev)
        mb = y.numel() * 2 / 1e6

        def run(fn, nbytes):
            t = timeit(lambda: (flush.zero_() if os.environ.get("FLUSH") else None, fn()))
            if os.environ.get("FLUSH"):
                t -= timeit(lambda: flush.zero_())
            return f"{nbytes / t / 1e9:6.0f}({t * 1e6:4.0f})"
        r = [run(lambda: ops.bn_act_apply(y, coef[0], coef[1], ACT_RELU, z, C, 0), 2 * mb * 1e6),
             run(lambda: ops.bn_act_apply(y, coef[0], coef[1], ACT_RELU, z, C, 0, zp), 2.25 * mb * 1e6),
             run(lambda: ops.bn_act_bwd_reduce(y, dz, C, 0, None, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part), 2 * mb * 1e6),
             run(lambda: ops.bn_act_bwd_apply(y, dz, C, 0, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy), 3 * mb * 1e6),
             run(lambda: ops.bn_act_bwd_reduce(y, dz, C, 0, dzp, coef[0], coef[1], coef[2], coef[3], ACT_RELU, part), 2.25 * mb * 1e6),
             run(lambda: ops.bn_act_bwd_apply(y, dz, C, 0, dzp, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dy), 3.25 * mb * 1e6)]
        print(f"{H:4d}^2 x {C:4d} {mb:9.1f} | {r[0]:>8s} {r[1]:>10s} {r[2]:>10s} {r[3]:>10s} {r[4]:>13s} {r[5]:>13s}")


if __name__ == "__main__":
    main()
