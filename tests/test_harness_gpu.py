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
