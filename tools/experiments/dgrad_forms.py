"""dgrad of the DoubleConv second convs (whose output feeds the first conv's BatchNorm backward): 8-wave vs 4-wave deferred form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
LAYERS = [("inc.3/u4.3", 256, 64), ("d1.3/u3.3", 128, 128), ("d2.3/u2.3", 64, 256), ("d3.3/u1.3", 32, 512)]
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda:0"); dt = torch.float16; N = 32
for name, H, C in LAYERS:
    dy = torch.randn(N, H, H, C, device=dev).to(dt); wd = (0.05 * torch.randn(9, C, C, device=dev)).to(dt)
    dx = torch.empty(N, H, H, C, device=dev, dtype=dt)
    r = {}
    for form in (8, 4):      # (form 0 = the register-staged kernel, printed as "staged")
        ops.conv3x3_set_kernel_form(form)
        r[form] = timeit(lambda: ops.conv3x3(dy, wd, dx, N, H, H, C, C, ops.TAPS3_DGRAD))
    ops.conv3x3_set_kernel_form(0)
    r[0] = timeit(lambda: ops.conv3x3(dy, wd, dx, N, H, H, C, C, ops.TAPS3_DGRAD))
    print(f"{name:12s} C={C:4d} @{H:3d}: staged {r[0]:7.1f} us | 8-wave {r[8]:7.1f} | 4-wave {r[4]:7.1f}", flush=True)
