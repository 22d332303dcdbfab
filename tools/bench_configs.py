#!/usr/bin/env python3
"""Timings of the non-headline BASELINE configs on one MI355X (informational; bench.py is the contract):
config 3: Generator / Discriminator / Unet training steps at 256x256 (batch 2 and 32);
config 5: UNet3D(1,2) forward+backward at 64^3 and 128^3 (batch 1); plus the eval-mode (inference) forward of the U-Net."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_segmentation_amd import steps  # noqa: E402
from semantic_segmentation_amd.harness import synthetic_batch  # noqa: E402
from semantic_segmentation_amd.losses import seg_loss  # noqa: E402
from semantic_segmentation_amd.models_pix2pix import networks  # noqa: E402
from semantic_segmentation_amd.unet import UNet  # noqa: E402
from semantic_segmentation_amd.unet3d import UNet3D  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def zero(*nets):
    for n in nets:
        for p in n.parameters():
            p.grad = None


def main():
    dev = torch.device("cuda:0")
    norm = networks.get_norm_layer("batch")
    G = networks.define_G(1, 1, 64, "unet_256", "batch", True).to(dev).train()
    D = networks.define_D(2, 64, "basic", 3, "batch").to(dev).train()
    U = UNet(1, 1).to(dev).train()
    networks.upconv_arch = (1e-3 * torch.randn(8, 3)).to(dev).requires_grad_(True)
    crit = networks.GANLoss("vanilla").to(dev)
    # pack reuse between the training forwards of one iteration: harness.EndToEndTrainer owns every update and sets
    # engine.trust_versions (version-keyed reuse); a foreign loop (Betty: `.data` writes) gets the safe default, where every
    # training forward re-packs (GSSEG_TRUST=0 times that)
    trust = os.environ.get("GSSEG_TRUST", "1") != "0"
    for m in (G, D, U):
        m.engine.trust_versions = trust
    print("pack reuse between training forwards:", "version-keyed (harness)" if trust else "none (safe default)")
    only3 = os.environ.get("GSSEG_BENCH_ONLY", "") == "config3"        # GSSEG_BENCH_ONLY=config3 [GSSEG_BENCH_B=2]: just the trio
    for B in ([int(os.environ["GSSEG_BENCH_B"])] if "GSSEG_BENCH_B" in os.environ else (2, 32)):
        x, mask = synthetic_batch(B, 256, seed=3)
        x, mask = x.to(dev), mask.to(dev)
        maskf = mask.float()
        real = torch.rand(B, 1, 256, 256, device=dev)

        def g_step():
            # the Generator step of iteration k+1 runs on the weights the Discriminator / Unet steps of iteration k already
            # packed for their forward passes: it re-uses that forward pack and builds the data-gradient pack (stale since
            # optimizer_G.step()).  The version bump that stands for optimizer_G.step() sits in d_step, so per trio the
            # merged packs are built exactly as often as in the real loop: one forward pack + one data-gradient pack.
            for key in [k for k in G._engine.packs._d if k[0] == "merged" and k[2]]:
                del G._engine.packs._d[key]                 # data-gradient packs: stale after optimizer_G.step()
            zero(G, D); steps.generator_step_loss(G, D, crit, maskf, real).backward()

        def d_step():
            # stand-in for optimizer_G.step() at the end of the Generator step: the first generator forward afterwards
            # (this one) pays the re-merge of the forward packs
            torch.autograd.graph.increment_version(list(G.parameters()))
            zero(G, D); steps.discriminator_step_loss(G, D, crit, maskf, real).backward()

        def u_step():
            zero(U); steps.unet_step_loss(U, G, x, mask, 1.0).backward()

        tg, td, tu = timeit(g_step, 10, 3), timeit(d_step, 10, 3), timeit(u_step, 10, 3)
        print(f"config3 B={B:3d}: generator step {tg * 1e3:8.2f} ms | discriminator step {td * 1e3:8.2f} ms | "
              f"unet step (2 U-Net fwd+bwd + G fwd + post-proc) {tu * 1e3:8.2f} ms | trio {B / (tg + td + tu):8.1f} img/s", flush=True)
        # the same three steps captured into hipGraphs and replayed (the launch path, ~16 us of Python / ctypes per launch, is what
        # bounds the eager numbers at batch 2).  Each graph contains the pack re-builds the eager emulation above pays per call:
        # captured in the order Discriminator (all generator packs stale: re-merged inside its graph), Unet (re-uses them; its own
        # packs re-built), Generator (re-uses the forward packs, re-merges the data-gradient packs) -- a capture only re-uses packs
        # that were built during a capture (pix2pix_engine._PackCache)
        if os.environ.get("GSSEG_BENCH_GRAPHS", "1") != "0":
            side = torch.cuda.Stream()
            pool, graphs = None, {}

            def body_g():
                zero(G, D); steps.generator_step_loss(G, D, crit, maskf, real).backward()

            def body_d():
                zero(G, D); steps.discriminator_step_loss(G, D, crit, maskf, real).backward()

            def body_u():
                zero(U); steps.unet_step_loss(U, G, x, mask, 1.0).backward()

            def stale_d():
                torch.autograd.graph.increment_version(list(G.parameters()))

            def stale_u():
                U.engine.invalidate_packs()

            def stale_g():
                for key in [k for k in G._engine.packs._d if k[0] == "merged" and k[2]]:
                    del G._engine.packs._d[key]

            from semantic_segmentation_amd.graphs import capture_step
            for name, body, stale in (("d", body_d, stale_d), ("u", body_u, stale_u), ("g", body_g, stale_g)):
                def prepare(stale=stale):
                    zero(G, D, U)
                    stale()
                graphs[name] = capture_step(body, prepare=prepare, stream=side, pool=pool)
                pool = pool or graphs[name].pool()
            with torch.cuda.stream(side):
                rg, rd, ru = (timeit(graphs[k].replay, 20, 3) for k in ("g", "d", "u"))
            print(f"config3 B={B:3d}: the same steps replayed from hipGraphs: generator {rg * 1e3:8.2f} ms | discriminator {rd * 1e3:8.2f} ms | "
                  f"unet {ru * 1e3:8.2f} ms | trio {B / (rg + rd + ru):8.1f} img/s", flush=True)
            del graphs, pool
    # the whole iteration through harness.EndToEndTrainer (three problems + their fused optimiser steps; no arch step, no
    # validation, no augmentation): the eager loop against forward + backward captured into hipGraphs
    from semantic_segmentation_amd.harness import EndToEndTrainer, SyntheticLungDataset
    for B in ([int(os.environ["GSSEG_BENCH_B"])] if "GSSEG_BENCH_B" in os.environ else (2,)):
        def mk(n, seed):                                 # batches resident on the device: the input pipeline is not what is timed
            ds = SyntheticLungDataset(n, 256, seed)
            return [{k: torch.stack([ds[i + j][k] for j in range(B)]).to(dev) for k in ("image", "mask")} for i in range(0, n, B)]
        for graphs in (False, True):
            tr = EndToEndTrainer(U, G, D, crit, mk(4 * B, 1), mk(B, 2), dev, unroll_steps=10 ** 9, valid_every=0, hip_graphs=graphs)
            tr.run(6, log_every=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.run(20, log_every=0)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 20
            print(f"config3 B={B:3d}: EndToEndTrainer iteration ({'hipGraphs' if graphs else 'eager    '}) {t * 1e3:8.2f} ms = {B / t:8.1f} img/s", flush=True)
            del tr
    if only3:
        return
    # config 1 (BASELINE configs[0], the reference's CPU-runnable case): UNet(1,2) 128x128 batch 4 forward + backward -- eager and
    # replayed from a hipGraph (launch-bound at this size)
    U1 = UNet(1, 2).to(dev).train()
    x1, m1 = synthetic_batch(4, 128, seed=9)
    x1, m1 = x1.to(dev), m1.to(dev)

    def c1_step():
        zero(U1)
        seg_loss(U1(x1), m1).backward()
    t_e = timeit(c1_step, 20, 3)
    from semantic_segmentation_amd.graphs import capture_step
    gr = capture_step(c1_step, prepare=lambda: zero(U1))
    with torch.cuda.stream(gr.stream):
        t_g = timeit(gr.replay, 20, 3)
    print(f"config1 UNet(1,2) 128x128 B=4 fwd+bwd: eager {t_e * 1e3:6.2f} ms = {4 / t_e:7.1f} img/s | hipGraph replay {t_g * 1e3:6.2f} ms = "
          f"{4 / t_g:7.1f} img/s", flush=True)
    del gr, U1
    # inference: eval-mode U-Net forward, BatchNorm folded into the convs vs the two-pass form
    from semantic_segmentation_amd.unet import unet_engine
    U2 = UNet(1, 2).to(dev).eval()
    xe, _ = synthetic_batch(32, 256, seed=5)
    xe = xe.to(dev)
    with torch.no_grad():
        for fold in (True, False):
            unet_engine.FOLD_BN_INFERENCE = fold
            t = timeit(lambda: U2(xe), iters=10, warm=3)
            print(f"inference UNet(1,2) 256x256 B=32 eval forward, BatchNorm {'folded into the convs' if fold else 'as a second pass':24s}: "
                  f"{t * 1e3:7.2f} ms = {32 / t:8.1f} img/s ({32 * 96.19e9 / t / 1e12:6.1f} TFLOP/s)")
    unet_engine.FOLD_BN_INFERENCE = True
    del G, D, U, U2
    torch.cuda.empty_cache()
    net = UNet3D(1, 2).to(dev).train()
    for S in (64, 128):
        xv = torch.randn(1, 1, S, S, S, device=dev)
        mv = (torch.rand(1, S, S, S, device=dev) > 0.5).long()

        def v_step():
            zero(net)
            lg = net(xv)
            n, c, d, h, w = lg.shape
            seg_loss(lg.reshape(n, c, d * h, w), mv.reshape(n, d * h, w)).backward()

        t = timeit(v_step, iters=3, warm=1)
        tf = 3 * 61.06e9 * (S / 32) ** 3 / t / 1e12
        print(f"config5 UNet3D(1,2) {S}^3 batch 1: fwd+bwd {t * 1e3:9.2f} ms  = {1 / t:7.2f} volumes/s  ({tf:6.1f} TFLOP/s algorithmic)")


if __name__ == "__main__":
    main()
