"""64->64 layer at 256^2: forward vs data gradient, with / without statistics, with / without resident weights (ablation bit 32)."""
import os, sys
os.environ["GSSEG_C3_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda:0"); dt = torch.float16; N, H, C = 32, 256, 64
x = torch.randn(N, H, H, C, device=dev).to(dt)
w = (0.05 * torch.randn(9, C, C, device=dev)).to(dt)
y = torch.empty(N, H, H, C, device=dev, dtype=dt)
part = torch.empty(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, C), C), device=dev)
for rep in range(2):
    for taps, tn in ((ops.TAPS3_FWD, "fwd taps"), (ops.TAPS3_DGRAD, "dgrad taps")):
        for p, pn in ((part, "stats"), (None, "no stats")):
            for act, an in ((0, "wres"), (32 << 8, "no wres")):
                t = timeit(lambda: ops.conv3x3(x, w, y, N, H, H, C, C, taps, None, p, act))
                print(f"{tn:10s} {pn:8s} {an:7s} {t:7.1f} us", flush=True)
