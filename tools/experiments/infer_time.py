"""eval-mode forward, UNet(1,2) 256^2 batch 32: default (pair) mode folded / two-pass, fast mode folded."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_segmentation_amd.unet import UNet, unet_engine
x = torch.randn(32, 1, 256, 256, device="cuda")
for name, precise, fold in (("default folded", None, True), ("default two-pass", None, False), ("fast folded", False, True)):
    torch.manual_seed(0)
    net = UNet(1, 2, precise=precise).cuda().eval()
    unet_engine.FOLD_BN_INFERENCE = fold
    with torch.no_grad():
        for _ in range(3): net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:18s} {ms:6.2f} ms = {32 / ms * 1e3:7.0f} img/s", flush=True)
unet_engine.FOLD_BN_INFERENCE = True
