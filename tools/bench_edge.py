"""Times the memory-bound edge kernels of the U-Net step at the headline shape (batch 32, 256x256, 64 channels):
stem forward / weight gradient (unet_parts.py:16 with one input channel) and the 1x1 head forward / backward
(unet_parts.py:74).  Usage on the GPU box: python tools/bench_edge.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from semantic_segmentation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 256
C = 64


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, 1, H, W, generator=g).to(dev)
w = (0.2 * torch.randn(C, 1, 3, 3, generator=g)).to(dev)
y = torch.empty(B, H, W, C, dtype=torch.float16, device=dev)
mt = ops.conv_smallcin_mtiles(B, H, W)
part = torch.empty(ops.bn_partials_numel(mt, C), dtype=torch.float32, device=dev)
dy = torch.randn(B, H, W, C, device=dev).half()
dw = torch.empty(C, 1, 3, 3, dtype=torch.float32, device=dev)
mb = 2.0 * B * H * W * C / 1e6
t = timeit(lambda: ops.conv_smallcin_fwd(x, w, None, y, part, 3, 1, 1))
print(f"stem fwd (+BN partials)   {t:7.1f} us  {mb / t:6.2f} TB/s of the {mb:.0f} MB output")
t = timeit(lambda: ops.conv_smallcin_fwd(x, w, None, y, None, 3, 1, 1))
print(f"stem fwd (no partials)    {t:7.1f} us  {mb / t:6.2f} TB/s")
t = timeit(lambda: ops.conv_smallcin_wgrad(x, dy, dw, 3, 1, 1, 1.0))
print(f"stem wgrad                {t:7.1f} us  {mb / t:6.2f} TB/s of the {mb:.0f} MB gradient read")
for ncls in (2, 1):
    wh = (0.1 * torch.randn(ncls, C, 1, 1, generator=g)).to(dev)
    bh = torch.zeros(ncls, device=dev)
    logits = torch.empty(B, ncls, H, W, dtype=torch.float32, device=dev)
    t = timeit(lambda: ops.conv_smallcout_fwd(dy, wh, bh, logits))
    print(f"head fwd   n_classes={ncls}    {t:7.1f} us  {mb / t:6.2f} TB/s")
    dl = torch.randn(B, ncls, H, W, device=dev)
    dz = torch.empty(B, H, W, C, dtype=torch.float16, device=dev)
    dwh = torch.empty_like(wh)
    dbh = torch.empty_like(bh)
    t = timeit(lambda: ops.conv_smallcout_bwd(dy, wh, dl, dz, dwh, dbh))
    print(f"head bwd   n_classes={ncls}    {t:7.1f} us  {2 * mb / t:6.2f} TB/s (activation read + gradient written)")
    t = timeit(lambda: ops.conv_smallcout_bwd(dy, wh, dl, None, dwh, dbh))
    print(f"  weight gradient only    {t:7.1f} us  {mb / t:6.2f} TB/s")
# stem backward: BatchNorm backward apply + weight gradient, as two kernels and as the fused pass
from semantic_segmentation_amd._lib import ACT_RELU  # noqa: E402
coef = torch.randn(4, C, device=dev).abs() + 0.5
c12 = 0.01 * torch.randn(2, C, device=dev)
dzt = torch.randn(B, H, W, C, device=dev).half()
dyt = torch.empty(B, H, W, C, dtype=torch.float16, device=dev)


def two():
    ops.bn_act_bwd_apply(y, dzt, C, 0, None, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, True, dyt)
    ops.conv_smallcin_wgrad(x, dyt, dw, 3, 1, 1, 1.0)


t = timeit(two)
print(f"stem bwd, two kernels     {t:7.1f} us  (apply: 2 reads + 1 write, wgrad: 1 read = {4 * mb:.0f} MB)")
t = timeit(lambda: ops.stem_bn_bwd_wgrad(y, dzt, C, 0, x, coef[0], coef[1], coef[2], coef[3], c12[0], c12[1], ACT_RELU, dw, 1.0))
print(f"stem bwd, fused           {t:7.1f} us  {2 * mb / t:6.2f} TB/s of the {2 * mb:.0f} MB read")
# the stem without its convolution output in memory
partm = torch.empty(ops.bn_partials_numel(mt, C), dtype=torch.float32, device=dev)
t = timeit(lambda: ops.stem_stats(x, w, partm))
print(f"stem statistics only      {t:7.1f} us")
t = timeit(lambda: ops.stem_fwd_bn(x, w, coef[0], coef[1], ACT_RELU, y))
print(f"stem fwd conv+BN+ReLU     {t:7.1f} us  {mb / t:6.2f} TB/s of the {mb:.0f} MB output")
partr = torch.empty(ops.bn_partials_numel(ops.bn_bwd_tiles(B, H, W), C), dtype=torch.float32, device=dev)
t = timeit(lambda: ops.bn_act_bwd_reduce(y, dzt, C, 0, None, coef[0], coef[1], coef[2], coef[3], ACT_RELU, partr))
print(f"bn bwd reduce (reads y, dz)  {t:7.1f} us")
t = timeit(lambda: ops.bn_act_apply(y, coef[0], coef[1], ACT_RELU, dyt, C, 0, None))
print(f"bn apply (y -> z)            {t:7.1f} us")
taps_ = torch.empty(mt * 54, dtype=torch.float32, device=dev)
ops.stem_stats(x, w, partm, taps_)
nts = ops.stem_bwd_tiles(B, H, W)
sws = torch.empty(nts * 576, dtype=torch.float32, device=dev)
s1p = torch.empty(nts * 64, dtype=torch.float32, device=dev)
t = timeit(lambda: ops.stem_bwd_onepass(x, y, dzt, C, 0, ACT_RELU, s1p, sws))
print(f"stem bwd one pass (z, dz)    {t:7.1f} us  {2 * mb / t:6.2f} TB/s of the {2 * mb:.0f} MB read")
dg, dbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
t = timeit(lambda: ops.stem_bwd_finalize(sws, s1p, taps_, w, coef[0], coef[2], coef[3], True, 1.0, dw, dg, dbt, B, H, W))
print(f"stem bwd finalize            {t:7.1f} us")
