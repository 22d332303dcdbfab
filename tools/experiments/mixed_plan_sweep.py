#!/usr/bin/env python3
"""Measured error / step time of candidate segment plans of the pair forward (GPU): max |dlogit| against two reference fixtures
and the 256^2 oracle case, and the bs=32 256^2 step time."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import numpy as np, torch
from oracle import oracle
from semantic_segmentation_amd.unet import UNet, unet_engine as ue
from semantic_segmentation_amd.losses import seg_loss
from semantic_segmentation_amd.harness import synthetic_batch
base = {s: ("xw" if s in ue.MIXED_XW else "1") for s in ue.STAGES}
plans = {"mixed": base,
         "up3 -> x": {**base, "up3.conv.0": "x", "up3.conv.3": "x"},
         "up3.c0 -> x": {**base, "up3.conv.0": "x"},
         "down2 -> x": {**base, "down2.0": "x", "down2.3": "x"},
         "up3, down2 -> x": {**base, "up3.conv.0": "x", "up3.conv.3": "x", "down2.0": "x", "down2.3": "x"}}
dev = torch.device("cuda:0")
cases = []
for name in ("unet_c2_128_b4", "unet_c1_64"):
    z = np.load(os.path.join(root, "tests", "golden", name + ".npz"))
    cases.append((name, int(z["n_classes"]), oracle.unet_state_dict(1, int(z["n_classes"]), seed=int(z["seed"])), torch.from_numpy(z["x"]), torch.from_numpy(z["logits"])))
sd = oracle.unet_state_dict(1, 2, seed=11)
x256, m256 = oracle.synthetic_batch(2, 256, seed=99)
cases.append(("oracle256", 2, sd, x256, oracle.unet_forward(sd, x256, True)))
xb, mb = synthetic_batch(32, 256, seed=1234)
xb, mb = xb.to(dev), mb.to(dev)
for pname, plan in plans.items():
    errs = []
    for name, ncls, sdc, x, ref in cases:
        net = UNet(1, ncls, precise=plan)
        net.load_state_dict(sdc, strict=True)
        net = net.to(dev).train()
        with torch.no_grad():
            d = (net(x.to(dev)).cpu() - ref).abs()
        errs.append(f"{name} {float(d.max()):.2e}")
    torch.manual_seed(1)
    net = UNet(1, 2, precise=plan).to(dev).train()
    def step():
        for p in net.parameters(): p.grad = None
        net.engine.invalidate_packs()
        seg_loss(net(xb), mb).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{pname:18s} {ms:7.2f} ms/step  " + "  ".join(errs), flush=True)
