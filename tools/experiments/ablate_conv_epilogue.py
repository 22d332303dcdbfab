#!/usr/bin/env python3
"""What the epilogue's global stores cost in conv3x3_dma_kernel (GSSEG_C3_DEBUG=1): normal / no global stores (dbg 64: pack,
LDS transpose, statistics all run) / no epilogue at all (dbg 4), interleaved rounds in one process."""
import os, sys, statistics
os.environ["GSSEG_C3_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
LAYERS = [("inc.3 fwd", 256, 64, 64, True), ("inc.3 nostat", 256, 64, 64, False), ("u4.0 fwd", 256, 128, 64, True), ("d1.3 fwd", 128, 128, 128, True),
          ("d2.3 fwd", 64, 256, 256, True), ("d3.3 fwd", 32, 512, 512, True)]
dev = torch.device("cuda:0"); dt = torch.float16; N = 32
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, H, Cin, Cout, stats in LAYERS:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt)
    w = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
    part = torch.empty(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout), device=dev) if stats else None
    res = {0: [], 64: [], 4: []}
    for rnd in range(4):
        for dbg in (0, 64, 4):
            res[dbg].append(timeit(lambda: ops.conv3x3(x, w, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part, act=(dbg << 8))))
    m = {k: statistics.median(v) for k, v in res.items()}
    print(f"{name:14s} normal {m[0]:7.1f} us  no stores {m[64]:7.1f} us ({100*(m[64]/m[0]-1):+5.1f} %)  no epilogue {m[4]:7.1f} us ({100*(m[4]/m[0]-1):+5.1f} %)", flush=True)
