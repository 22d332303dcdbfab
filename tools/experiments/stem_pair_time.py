"""stem pair kernel: LDS-staged full-line stores (rows256) vs lane-by-lane pieces; same bits, time each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_segmentation_amd import ops
from semantic_segmentation_amd._lib import ACT_RELU
torch.manual_seed(0)
dev = "cuda"
N, H, W = 32, 256, 256
x = torch.randn(N, 1, H, W, device=dev)
w = torch.randn(64, 1, 3, 3, device=dev) * 0.3
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.1
out = {}
for q in (0, 1):
    for off in (1, 0):
        if off: os.environ["GSSEG_STEM_ROWS256_OFF"] = "1"
        else: os.environ.pop("GSSEG_STEM_ROWS256_OFF", None)
        z = torch.zeros(N, H, W, 128, device=dev, dtype=torch.float16)
        f = (lambda: ops.stem_fwd_bn_pair_q8(x, w, sc, sh, ACT_RELU, z)) if q else (lambda: ops.stem_fwd_bn_pair(x, w, sc, sh, ACT_RELU, z))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        out[(q, off)] = z.clone()
        print(f"q8={q} rows256={'off' if off else 'on '}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  ({z.numel() * 2 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:.2f} TB/s)", flush=True)
    print("   identical:", torch.equal(out[(q, 0)].view(torch.int16), out[(q, 1)].view(torch.int16)), flush=True)
# ragged: M not a multiple of 256
x2 = torch.randn(3, 1, 70, 52, device=dev)
zs = []
for off in (1, 0):
    if off: os.environ["GSSEG_STEM_ROWS256_OFF"] = "1"
    else: os.environ.pop("GSSEG_STEM_ROWS256_OFF", None)
    z = torch.zeros(3, 70, 52, 128, device=dev, dtype=torch.float16)
    ops.stem_fwd_bn_pair(x2, w, sc, sh, ACT_RELU, z); torch.cuda.synchronize(); zs.append(z)
print("ragged identical:", torch.equal(zs[0].view(torch.int16), zs[1].view(torch.int16)))
