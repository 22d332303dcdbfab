"""per-layer conv3x3 weight-gradient time (slabs + ordered reduce) for the current GSSEG_W3_DMA setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_segmentation_amd import ops
LAYERS = [("inc.3", 256, 64, 64), ("d1.0", 128, 64, 128), ("d1.3", 128, 128, 128), ("d2.0", 64, 128, 256),
          ("d2.3", 64, 256, 256), ("d3.0", 32, 256, 512), ("d3.3", 32, 512, 512), ("d4.0", 16, 512, 1024),
          ("d4.3", 16, 1024, 1024), ("u1.0", 32, 1024, 512), ("u2.0", 64, 512, 256), ("u3.0", 128, 256, 128), ("u4.0", 256, 128, 64)]
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
dev = torch.device("cuda:0"); dt = torch.float16; N = int(os.environ.get("NB", "32"))
tot = 0.0
print("W3_DMA =", os.environ.get("GSSEG_W3_DMA", "1"))
for name, H, Cin, Cout in LAYERS:
    x = torch.randn(N, H, H, Cin, device=dev).to(dt); dy = torch.randn(N, H, H, Cout, device=dev).to(dt)
    ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, H, H, Cin, Cout), dtype=torch.float32, device=dev)
    g = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev)
    fl = 2.0 * N * H * H * Cout * Cin * 9
    t = timeit(lambda: ops.conv3x3_wgrad_det(x, dy, ws, g, N, H, H, Cin, Cout, 1.0))
    tot += t
    print(f"{name:6s} wgrad+reduce {fl/t/1e12:7.1f} TF  ({t*1e6:6.1f} us)  parts {ops.conv3x3_wgrad_parts(N, H, H, Cin, Cout)}", flush=True)
print("sum ms %.3f" % (tot * 1e3))
