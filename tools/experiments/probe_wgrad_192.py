"""2-D weight gradient (the per-depth-tap launch of the 3-D net's s_block1.conv1: 128 slices of 128x128, 192 -> 64) in variants:
dense 192-channel input, 64-channel slices of a 192-wide buffer, other Cin / sizes -- to locate the 0.72 PF of that layer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
dev = torch.device("cuda:0"); dt = torch.float16
def run(tag, N, H, Cin, Cout, stride=None, coff=0):
    st = Cin if stride is None else stride
    x = torch.randn(N, H, H, st, device=dev).to(dt); dy = torch.randn(N, H, H, Cout, device=dev).to(dt)
    ws = torch.empty(ops.conv3x3_wgrad_ws_floats(N, H, H, Cin, Cout), dtype=torch.float32, device=dev)
    g = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev)
    fl = 2.0 * N * H * H * Cout * Cin * 9
    t = timeit(lambda: ops.conv3x3_wgrad_det(x, dy, ws, g, N, H, H, Cin, Cout, 1.0, in_stride=st, in_coff=coff))
    print(f"{tag:34s} N{N:4d} H{H:4d} {Cin:4d}->{Cout:4d} stride {st:4d}: {fl/t/1e12:7.1f} TF ({t*1e6:7.1f} us) parts {ops.conv3x3_wgrad_parts(N, H, H, Cin, Cout)}", flush=True)
run("3-D s1.conv1 depth tap", 128, 128, 192, 64)
run("same pixels as 32 x 256^2", 32, 256, 192, 64)
run("128 -> 64 (2-D up4.conv.0)", 32, 256, 128, 64)
run("128 -> 64, 128 slices of 128^2", 128, 128, 128, 64)
run("64 -> 64, 128 x 128^2", 128, 128, 64, 64)
run("64 of a 192-wide buffer", 128, 128, 64, 64, stride=192, coff=64)
run("256 -> 64", 128, 128, 256, 64)
run("192 -> 128", 128, 128, 192, 128)
run("384 -> 128 @64 (s2.conv1 tap)", 64, 64, 384, 128)
run("256 -> 128 @64", 64, 64, 256, 128)
