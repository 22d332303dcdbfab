"""per-level time of the U-Net k2/s2 transposed conv: forward, data gradient, weight gradient (+ bias column sum)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_segmentation_amd import ops
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
dev = torch.device("cuda:0"); dt = torch.float16; N = 32
tot = [0.0, 0.0, 0.0]
for name, h, cin in [("up1", 16, 1024), ("up2", 32, 512), ("up3", 64, 256), ("up4", 128, 128)]:
    cout = cin // 2; H2 = 2 * h
    x = torch.randn(N, h, h, cin, device=dev).to(dt)
    w = (0.05 * torch.randn(cin, cout, 2, 2, device=dev))
    wf = torch.empty(4, cout, cin, dtype=dt, device=dev); wd = torch.empty(4, cin, cout, dtype=dt, device=dev)
    ops.pack_weight(w, wf, wd, True)
    bias = torch.zeros(cout, device=dev)
    cat = torch.zeros(N, H2, H2, 2 * cout, dtype=dt, device=dev)
    dcat = torch.randn(N, H2, H2, 2 * cout, device=dev).to(dt)
    taps = [(py, px) for py in range(2) for px in range(2)]
    g = ops.make_geom(N, H2, H2, cout, h, h, cin, h, h, taps, isy=2, isx=2, in_stride=2 * cout, in_coff=cout)
    dz = torch.empty(N, h, h, cin, dtype=dt, device=dev)
    dw = torch.empty(cin, cout, 2, 2, device=dev)
    ws = torch.empty(ops.conv_wgrad_ws_floats(g), device=dev)
    fl = 2.0 * N * h * h * cin * 4 * cout
    tf = timeit(lambda: ops.upconv2x2_fwd(x, wf, bias, cat, N, 1, h, h, cin, cout, 1, H2, H2, out_stride=2 * cout, out_coff=cout))
    td = timeit(lambda: ops.upconv2x2_dgrad(g, dcat, wd, dz, N, h, h, cin, cout, H2, H2, 2 * cout, cout, 0, 0))
    tw = timeit(lambda: ops.conv_wgrad_det(g, dcat, x, ws, dw, cin, cout, 4, 1.0))
    for i, t in enumerate((tf, td, tw)): tot[i] += t
    print(f"{name} {cin}->{cout} @{h}: fwd {tf*1e6:6.1f} us ({fl/tf/1e12:5.0f} TF)  dgrad {td*1e6:6.1f} us ({fl/td/1e12:5.0f} TF)  wgrad {tw*1e6:6.1f} us ({fl/tw/1e12:5.0f} TF)", flush=True)
print("sum us: fwd %.0f dgrad %.0f wgrad %.0f" % tuple(t * 1e6 for t in tot))
