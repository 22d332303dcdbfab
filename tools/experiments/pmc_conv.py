#!/usr/bin/env python3
"""Run a few isolated conv launches (for rocprofv3 --pmc collection): python tools/pmc_conv.py [layer]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_segmentation_amd import ops
dev = torch.device("cuda:0"); dt = torch.float16; N = 32
LAY = {"inc.3": (256, 64, 64), "d2.3": (64, 256, 256), "d4.3": (16, 1024, 1024), "u4.0": (256, 128, 64)}
for name in (sys.argv[1:] or ["d2.3"]):
    H, Cin, Cout = LAY[name]
    x = torch.randn(N, H, H, Cin, device=dev).to(dt); dy = torch.randn(N, H, H, Cout, device=dev).to(dt)
    wf = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt); dw = torch.zeros(9, Cout, Cin, device=dev)
    part = torch.empty(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout), device=dev)
    for _ in range(3):
        ops.conv3x3(x, wf, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part)
        ops.conv3x3_wgrad(x, dy, dw, N, H, H, Cin, Cout)
    torch.cuda.synchronize()
