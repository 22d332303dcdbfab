"""ablation timing of the conv3x3 kernel on selected layers (GSSEG_C3_DEBUG=1)."""
import os, sys
os.environ["GSSEG_C3_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
LAYERS = [("inc.3", 256, 64, 64), ("d1.3", 128, 128, 128), ("u4.0", 256, 128, 64), ("d2.3", 64, 256, 256), ("u2.0", 64, 512, 256), ("d3.3", 32, 512, 512), ("u1.0", 32, 1024, 512)]
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
dev = torch.device("cuda:0"); dt = torch.float16; N = 32
print("DMA =", os.environ.get("GSSEG_C3_DMA", "0"))
for name, H, Cin, Cout in LAYERS:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt)
    wf = (0.05 * torch.randn(9, Cout, Cin, device=dev)).to(dt)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=dt)
    part = torch.empty(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, Cout), Cout), device=dev)
    res = []
    for bits in ([0] * 6 if os.environ.get("ABL_FULL_ONLY") else (0, 0x800, 0x1000, 0x1800, 0x200, 0x600)):
        t = timeit(lambda: ops.conv3x3(x, wf, y, N, H, H, Cin, Cout, ops.TAPS3_FWD, None, part, act=bits))
        res.append(t * 1e6)
    print(f"{name:6s} full {res[0]:6.1f}  no-weight-dma {res[1]:6.1f}  no-halo-dma {res[2]:6.1f}  no-dma {res[3]:6.1f}  no-mfma {res[4]:6.1f}  dma-only {res[5]:6.1f} us", flush=True)
