#!/usr/bin/env python3
"""In-kernel phase cycles of conv3x3_dma_kernel: a diagnostic library built with -DGS_C3_PHASE_TIMING (tools/build_phase_lib.sh)
is selected through GSSEG_LIB; block 0 writes each wave's per-phase cycle sums behind the BatchNorm partial rows."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GSSEG_LIB"] = os.path.join(root, "semantic_segmentation_amd", "libgsseg_hip_phase.so")
sys.path.insert(0, root)
import torch
from semantic_segmentation_amd import ops
dev = torch.device('cuda:0'); dt = torch.float16; N = 32
names = ["sync even", "stage even", "sync odd", "stage odd", "boundary (convert)", "item head"]
for name, H, Cin, Cout in [("inc.3 64->64@256", 256, 64, 64), ("u4.0 128->64@256", 256, 128, 64), ("d1.3 128->128@128", 128, 128, 128),
                           ("d2.3 256->256@64", 64, 256, 256), ("d3.3 512->512@32", 32, 512, 512)]:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt); wf = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
    part = torch.zeros(max(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout), 301 * 2 * Cout + 64), device=dev)
    for _ in range(3):
        ops.conv3x3(x, wf, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part)
    torch.cuda.synchronize()
    allw = part[300 * 2 * Cout: 300 * 2 * Cout + 64].cpu().view(8, 8)
    for w in (0, 5):
        ph = allw[w].tolist()[:6]; tot = sum(ph)
        print(f"{name:20s} wave{w} cycles {int(tot):8d}  " + "  ".join(f"{n} {100*v/tot:4.1f}%" for n, v in zip(names, ph)), flush=True)
