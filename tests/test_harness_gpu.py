"""Betty-free end-to-end harness (semantic_segmentation_amd/harness.py): the four problems of
running_files/train_end2end_jsrt.py step on the HIP engines with the fused optimisers, validation saves the best
U-Net and the loss of a tiny over-fit run goes down."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_end_to_end_iterations(tmp_path):
    from semantic_segmentation_amd.harness import EndToEndTrainer, SyntheticLungDataset
    from semantic_segmentation_amd.models_pix2pix import networks
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = UNet(1, 1).to(dev)
    netG = networks.define_G(1, 1, 64, "unet_256", "batch", True, "normal", 0.02, [0])
    netD = networks.define_D(2, 64, "basic", 3, "batch", "normal", 0.02, [0])
    crit = networks.GANLoss("vanilla").to(dev)
    mk = lambda n, seed: torch.utils.data.DataLoader(SyntheticLungDataset(n, 256, seed), batch_size=2, shuffle=False, drop_last=True)
    tr = EndToEndTrainer(net, netG, netD, crit, mk(4, 1), mk(2, 2), dev, unet_lr=1e-4, unroll_steps=2, valid_every=3,
                         save_dir=str(tmp_path))
    w0 = {k: v.detach().clone() for k, v in list(net.state_dict().items())[:4]}
    g0 = next(netG.parameters()).detach().clone()
    d0 = next(netD.parameters()).detach().clone()
    tr.run(6, log_every=0)
    assert tr.global_step == 6 and len(tr.history) == 6
    for rec in tr.history:
        assert all(torch.isfinite(torch.tensor(v)) for v in rec.values()), rec
    assert "loss_arch" in tr.history[1] and "val_score" in tr.history[2]
    assert 0.0 <= tr.val_best_score <= 1.0
    assert os.path.exists(tmp_path / "unet.pkl") and os.path.exists(tmp_path / "final.pkl")
    assert any(not torch.equal(w0[k], net.state_dict()[k]) for k in w0)
    assert not torch.equal(g0, next(netG.parameters()).detach()) and not torch.equal(d0, next(netD.parameters()).detach())
    # the saved checkpoint loads into a fresh reference-compatible module
    UNet(1, 1).load_state_dict(torch.load(tmp_path / "final.pkl", map_location="cpu"), strict=True)


@pytest.mark.parametrize("unroll_steps,iters", [(3, 7), (2, 8)])
def test_hip_graph_iterations_match_eager(unroll_steps, iters):
    """(unroll_steps = 2: NO arch step falls between the capture of the Discriminator graph and the capture of the Generator graph
    -- (graph_warmup + 1) % unroll_steps != 0 -- so the Generator graph holds no merge kernel of its own and reads the forward packs
    the Discriminator graph merges; the eager arch steps in between must refresh them, with an arch learning rate
    large enough for a stale pack to show.)
    hip_graphs=True: forward + backward of the Generator / Discriminator / Unet problems captured into hipGraphs after two eager
    iterations and replayed (static inputs, eager optimiser / arch steps in between, one validation) -- the same losses and the
    same weights as the eager loop on the same data (no dropout in G, so the two runs draw no random numbers)."""
    from semantic_segmentation_amd.harness import EndToEndTrainer, SyntheticLungDataset
    from semantic_segmentation_amd.models_pix2pix import networks
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    mk = lambda n, seed: torch.utils.data.DataLoader(SyntheticLungDataset(n, 256, seed), batch_size=2, shuffle=False, drop_last=True)

    def run(graphs):
        torch.manual_seed(0)
        networks.upconv_arch.data.zero_()
        net = UNet(1, 1).to(dev)
        netG = networks.define_G(1, 1, 64, "unet_256", "batch", False, "normal", 0.02, [0])
        netD = networks.define_D(2, 64, "basic", 3, "batch", "normal", 0.02, [0])
        crit = networks.GANLoss("vanilla").to(dev)
        tr = EndToEndTrainer(net, netG, netD, crit, mk(6, 1), mk(2, 2), dev, unet_lr=1e-4, unroll_steps=unroll_steps, valid_every=5,
                             arch_through_generator=True, hip_graphs=graphs, arch_lr=0.3 if unroll_steps == 2 else 3e-4)
        tr.run(iters, log_every=0)
        torch.cuda.synchronize()
        if graphs:
            assert set(tr._graphs) == {"G", "D", "U"}
        hist = [{k: float(v) for k, v in rec.items()} for rec in tr.history]
        return hist, [p.detach().clone() for m in (net, netG, netD) for p in m.parameters()], \
            [b.detach().clone() for m in (net, netG, netD) for b in m.buffers()]

    h0, p0, b0 = run(False)
    h1, p1, b1 = run(True)
    assert len(h0) == len(h1) == iters
    for r0, r1 in zip(h0, h1):
        assert r0.keys() == r1.keys()
        for k in r0:
            assert abs(r0[k] - r1[k]) <= 1e-5 * max(1.0, abs(r0[k])), (k, r0[k], r1[k])
    worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(p0, p1))
    assert worst < 1e-4, worst
    for a, b in zip(b0, b1):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-6)


def test_hip_graphs_with_dropout_and_augmentation():
    """the stock configuration under hip_graphs: dropout in the generator (random numbers drawn INSIDE the captured steps: every
    replay must draw fresh ones), the host-side mask augmentation outside the graphs, validation and arch steps in between"""
    from semantic_segmentation_amd.augment import MaskAugmenter
    from semantic_segmentation_amd.harness import EndToEndTrainer, SyntheticLungDataset
    from semantic_segmentation_amd.models_pix2pix import networks
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = UNet(1, 1).to(dev)
    netG = networks.define_G(1, 1, 64, "unet_256", "batch", True, "normal", 0.02, [0])
    netD = networks.define_D(2, 64, "basic", 3, "batch", "normal", 0.02, [0])
    crit = networks.GANLoss("vanilla").to(dev)
    mk = lambda n, seed: torch.utils.data.DataLoader(SyntheticLungDataset(n, 256, seed), batch_size=2, shuffle=False, drop_last=True)
    tr = EndToEndTrainer(net, netG, netD, crit, mk(4, 1), mk(2, 2), dev, unet_lr=1e-4, unroll_steps=2, valid_every=4,
                         mask_augment=MaskAugmenter(seed=0), hip_graphs=True)
    tr.run(8, log_every=0)
    torch.cuda.synchronize()
    assert set(tr._graphs) == {"G", "D", "U"}
    for rec in tr.history:
        assert all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in rec.values()), rec
    # the same batch comes back every second iteration (4 samples, batch 2): with fresh dropout masks and updated weights the
    # generator loss of two replays on the same batch differs
    lg = [float(rec["loss_G"]) for rec in tr.history]
    assert lg[4] != lg[6] and lg[5] != lg[7], lg
    assert 0.0 <= tr.val_best_score <= 1.0


def test_capture_step_unet_matches_eager():
    """graphs.capture_step on a plain U-Net step (static inputs, gradients set to None and packs dropped before the capture): every
    replay reproduces the eager step bit for bit -- also after the batch in the static tensors and the weights have changed"""
    from semantic_segmentation_amd.graphs import capture_step
    from semantic_segmentation_amd.harness import synthetic_batch
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = UNet(1, 2).to(dev).train()
    net.engine.trust_versions = True
    xs, ms = [], []
    for seed in (1, 2):
        x, m = synthetic_batch(2, 64, seed=seed)
        xs.append(x.to(dev)); ms.append(m.to(dev))
    x_st, m_st = xs[0].clone(), ms[0].clone()

    def step():
        loss = seg_loss(net(x_st), m_st)
        loss.backward()
        return loss.detach()

    def prepare():
        for p in net.parameters():
            p.grad = None
        net.engine.invalidate_packs()

    def eager(i):
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        x_st.copy_(xs[i]); m_st.copy_(ms[i])
        prepare()
        loss = float(step())
        grads = [p.grad.clone() for p in net.parameters()]
        net.load_state_dict(sd)                  # undo the BatchNorm running-statistics update
        return loss, grads

    cap = capture_step(step, prepare=prepare)
    for i in (1, 0, 1):
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.0 + 0.01 * (i + 1))     # weights change between replays: the packs are re-built inside the graph
        l_ref, g_ref = eager(i)
        x_st.copy_(xs[i]); m_st.copy_(ms[i])
        with torch.cuda.stream(cap.stream):
            l_got = float(cap.replay())
        torch.cuda.synchronize()
        assert l_got == l_ref, (i, l_got, l_ref)
        for p, g in zip(net.parameters(), g_ref):
            assert torch.equal(p.grad, g)


def test_unet_only_loss_decreases():
    """over-fit 2 synthetic batches with the fused RMSprop: the segmentation loss must drop"""
    from semantic_segmentation_amd.harness import SyntheticLungDataset
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.optim import RMSprop
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = UNet(1, 2).to(dev)
    opt = RMSprop(net.parameters(), lr=1e-4, weight_decay=1e-8, momentum=0.9)
    ds = SyntheticLungDataset(4, 128, 3)
    x = torch.stack([ds[i]["image"] for i in range(4)]).to(dev)
    m = torch.stack([ds[i]["mask"] for i in range(4)]).to(dev)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = seg_loss(net(x), m)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.8 * losses[0], losses
