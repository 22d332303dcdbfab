#!/usr/bin/env python3
"""One training step of a non-headline BASELINE config in a loop, for rocprofv3 (tools/profile_configs.sh).
  g<B> / d<B> / u<B> : Generator / Discriminator / Unet step of config 3 at batch B (train_end2end_jsrt.py:141-226), with the
                       once-per-iteration weight change of the trio emulated as tools/bench_configs.py does
  v<S>               : UNet3D(1,2) forward+backward at S^3, batch 1 (config 5)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_segmentation_amd import steps  # noqa: E402
from semantic_segmentation_amd.harness import synthetic_batch  # noqa: E402
from semantic_segmentation_amd.losses import seg_loss  # noqa: E402
from semantic_segmentation_amd.models_pix2pix import networks  # noqa: E402

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
warm = 2
dev = torch.device("cuda:0")
kind, size = which[0], int(which[1:])
if kind in "gdu":
    from semantic_segmentation_amd.unet import UNet
    B = size
    G = networks.define_G(1, 1, 64, "unet_256", "batch", True).to(dev).train()
    D = networks.define_D(2, 64, "basic", 3, "batch").to(dev).train()
    U = UNet(1, 1).to(dev).train()
    networks.upconv_arch = (1e-3 * torch.randn(8, 3)).to(dev).requires_grad_(True)
    crit = networks.GANLoss("vanilla").to(dev)
    for m_ in (G, D, U):
        m_.engine.trust_versions = True          # as harness.EndToEndTrainer (it owns every update): version-keyed pack reuse
    x, mask = synthetic_batch(B, 256, seed=3)
    x, mask = x.to(dev), mask.to(dev)
    maskf, real = mask.float(), torch.rand(B, 1, 256, 256, device=dev)

    def zero(*nets):
        for n in nets:
            for p in n.parameters():
                p.grad = None

    def step():
        if kind == "g":
            # as in the loop: the forward packs of the previous Discriminator step are re-used, the data-gradient packs are
            # stale (weights changed by optimizer_G.step()) -- emulated by dropping only those
            for key in [k for k in G._engine.packs._d if k[0] == "merged" and k[2]]:
                del G._engine.packs._d[key]
            zero(G, D); steps.generator_step_loss(G, D, crit, maskf, real).backward()
        elif kind == "d":
            torch.autograd.graph.increment_version(list(G.parameters()))
            zero(G, D); steps.discriminator_step_loss(G, D, crit, maskf, real).backward()
        else:
            zero(U); steps.unet_step_loss(U, G, x, mask, 1.0).backward()
else:
    from semantic_segmentation_amd.unet3d import UNet3D
    net = UNet3D(1, 2).to(dev).train()
    S = size
    xv = torch.randn(1, 1, S, S, S, device=dev)
    mv = (torch.rand(1, S, S, S, device=dev) > 0.5).long()

    def step():
        for p in net.parameters():
            p.grad = None
        lg = net(xv)
        n, c, d, h, w = lg.shape
        seg_loss(lg.reshape(n, c, d * h, w), mv.reshape(n, d * h, w)).backward()

for _ in range(warm):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print(which, "ms/step", round((time.perf_counter() - t) / iters * 1e3, 3), "steps", iters + warm, flush=True)
