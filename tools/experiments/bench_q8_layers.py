import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from semantic_segmentation_amd import ops
dev = torch.device("cuda:0")
shapes = [(32, 256, 256, 64, 64), (32, 256, 256, 128, 64), (32, 128, 128, 64, 128), (32, 128, 128, 128, 128), (32, 128, 128, 256, 128),
          (32, 64, 64, 128, 256), (32, 64, 64, 256, 256)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, cin, cout) in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(N, H, W, 2 * cin, device=dev, generator=g).abs()).half()
    # q-plane: make it valid e4m3 bytes by running the producer
    yh = x[..., :cin].contiguous(); yl = (torch.randn(N, H, W, cin, device=dev, generator=g) * 1e-4).half()
    one = torch.ones(cin, device=dev); zero = torch.zeros(cin, device=dev)
    xq = torch.empty(N, H, W, 2 * cin, dtype=torch.float16, device=dev)
    ops.bn_act_apply_split_q8(yh, yl, one, zero, 0, xq, xq[..., cin:], True, 2 * cin, 0)
    xp = torch.empty(N, H, W, 2 * cin, dtype=torch.float16, device=dev)
    ops.bn_act_apply_split(yh, yl, one, zero, 0, xp, xp[..., cin:], 2 * cin, 0)
    wt = ((torch.rand(cout, cin, 3, 3, device=dev, generator=g) * 2 - 1) / (cin * 9) ** 0.5).contiguous()
    qpack = torch.empty(9, cout, 2 * cin, dtype=torch.float16, device=dev); wexp = torch.empty(cout, dtype=torch.int32, device=dev)
    ops.pack_weight_q8([(wt, qpack, wexp)])
    spack = torch.empty(9, cout, 3 * cin, dtype=torch.float16, device=dev)
    ops.pack_weight_segs([(wt, spack, False, [(0, 0, cin), (0, 0, cin), (1, 0, cin)])])
    w1 = torch.empty(9, cout, cin, dtype=torch.float16, device=dev)
    ops.pack_weight(wt, w1, None, False)
    y_hi = torch.empty(N, H, W, cout, dtype=torch.float16, device=dev); y_lo = torch.empty_like(y_hi)
    def parts(K, q=False): return torch.zeros(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, W, cout), cout), dtype=torch.float32, device=dev)
    pq, px, p1 = parts(2 * cin), parts(3 * cin), torch.zeros(ops.bn_partials_numel(ops.conv3x3_stat_rows(N, H, W, cin, cout), cout), dtype=torch.float32, device=dev)
    tq = timeit(lambda: ops.conv3x3_q8(xq, qpack, wexp, y_hi, y_lo, N, H, W, cin, cout, 2 * cin, 0, pq))
    ops.conv3x3_set_kernel_form(4)
    pq4 = parts(2 * cin); px4 = parts(3 * cin)
    tq4 = timeit(lambda: ops.conv3x3_q8(xq, qpack, wexp, y_hi, y_lo, N, H, W, cin, cout, 2 * cin, 0, pq4))
    tx4 = timeit(lambda: ops.conv3x3_segs(xp, spack, y_hi, y_lo, N, H, W, 3 * cin, 2 * cin, cin, cout, in_stride=2 * cin, bn_partials=px4))
    ops.conv3x3_set_kernel_form(-1)
    tx = timeit(lambda: ops.conv3x3_segs(xp, spack, y_hi, y_lo, N, H, W, 3 * cin, 2 * cin, cin, cout, in_stride=2 * cin, bn_partials=px))
    t1p = timeit(lambda: ops.conv3x3_segs(xp, w1, y_hi, y_lo, N, H, W, cin, cin, cin, cout, in_stride=2 * cin, bn_partials=parts(cin)))
    t1 = timeit(lambda: ops.conv3x3(xp, w1, y_hi, N, H, W, cin, cout, ops.TAPS3_FWD, None, p1, in_stride=2 * cin))
    gf = 2.0 * N * H * W * cout * 9 * cin / 1e9
    print(f"{cin:4d}->{cout:4d} @{H:3d}: default {t1:6.1f} us ({gf/t1/1e3:.2f} PF)  pair'1' {t1p:6.1f}  xw {tx:6.1f} ({3*gf/tx/1e3:.2f} PF exec)  q {tq:6.1f} ({2*gf/tq/1e3:.2f} PF-eq)  q/xw {tq/tx:.2f} | 4-wave: xw {tx4:6.1f} q {tq4:6.1f}", flush=True)
