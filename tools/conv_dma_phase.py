"""in-kernel phase cycles of conv3x3_dma_kernel (library built with -DGS_C3_PHASE_TIMING; GSSEG_LIB points at it)."""
import sys, os, torch
sys.path.insert(0, '.')
from semantic_segmentation_amd import ops
dev = torch.device('cuda:0'); dt = torch.float16; N = 32
names = ["sync even", "stage even", "sync odd", "setup+stage odd", "pre-epi barrier", "epilogue+zero", "post syncthreads", "stats+advance"]
for name, H, Cin, Cout in [("inc.3", 256, 64, 64), ("u4.0", 256, 128, 64), ("d1.3", 128, 128, 128), ("u2.0", 64, 512, 256)]:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt); wf = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
    part = torch.zeros(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout), device=dev)
    for _ in range(3):
        ops.conv3x3(x, wf, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part)
    torch.cuda.synchronize()
    allw = part[:64].cpu().view(8, 8)
    for w in (0, 5):
        ph = allw[w].tolist(); tot = sum(ph)
        print(name, f"wave{w} cycles {int(tot)}", " ".join(f"{n}={v:.0f}({100*v/tot:.0f}%)" for n, v in zip(names, ph)), flush=True)
