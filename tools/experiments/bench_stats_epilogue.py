"""What the BatchNorm statistics epilogue of gs_conv3x3 costs per layer shape: the same launch without and with bn_partials,
minimum / median of four runs of 60 launches, alternating (box noise is +-4 %).  Usage on the GPU box:
python tools/bench_stats_epilogue.py   (GSSEG_LIB=<alternate .so> compares builds on the same box)"""
import sys, torch
sys.path.insert(0, ".")
from semantic_segmentation_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=60):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = 32
for (H, cin, cout, taps) in [(256, 64, 128, ops.TAPS3_DGRAD), (256, 64, 64, ops.TAPS3_FWD), (256, 128, 64, ops.TAPS3_FWD), (128, 128, 128, ops.TAPS3_FWD)]:
    x = torch.randn(N, H, H, cin, device=dev).half()
    w = (0.05 * torch.randn(9, cout, cin, device=dev)).half()
    y = torch.empty(N, H, H, cout, dtype=torch.float16, device=dev)
    part = torch.empty(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, cout), cout), dtype=torch.float32, device=dev)
    fp = lambda: ops.conv3x3(x, w, y, N, H, H, cin, cout, taps)
    fs = lambda: ops.conv3x3(x, w, y, N, H, H, cin, cout, taps, bn_partials=part)
    tp, ts = [], []
    for _ in range(4):
        tp.append(timeit(fp)); ts.append(timeit(fs))
    print(f"H={H} {cin}->{cout}: plain min {min(tp):7.1f} med {sorted(tp)[2]:7.1f} us, with partials min {min(ts):7.1f} med {sorted(ts)[2]:7.1f} us  ({(min(ts) / min(tp) - 1) * 100:+.1f} %)")
