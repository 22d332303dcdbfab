#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE
modules (read-only checkout at /root/reference) on CPU, in the build container.

The reference never travels to the GPU box: only the .npz files written here do.
Large tensors (the 31 M-parameter U-Net weights, the 272 M-parameter generator
weights) are NOT stored; they are regenerated from a seed by
``oracle.unet_state_dict`` / ``golden_util.seeded_*`` (our own code) and loaded
into the reference modules with ``load_state_dict(strict=True)`` -- which also
checks that our key names and shapes are exactly the reference's.  A checksum
of every generated tensor is stored so a drift of the PRNG stream is detected.

Usage:  python tests/golden/make_golden.py [--only unet|unet_b32|bilinear|ops|jaccard|pix2pix|pix2pix_b8|unet3d|unet3d_128]
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True          # the reference checkout is read-only: no __pycache__ there

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("GENSEG_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from golden_util import (grad_summary, seeded_discriminator_state_dict,  # noqa: E402
                         seeded_generator_state_dict, tensor_checksum)
from oracle import oracle  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def ref_seg_loss(logits, mask, dice_loss):
    """train_end2end_jsrt.py:181-183 (classes=1) and the canonical multiclass form."""
    import torch.nn.functional as F
    n_classes = logits.shape[1]
    if n_classes == 1:
        t = mask.float()
        bce = torch.nn.BCEWithLogitsLoss()(logits, t)
        dl = dice_loss(torch.sigmoid(logits.squeeze(1)), t.squeeze(1))
        return bce, dl
    m = mask[:, 0].long()
    ce = torch.nn.CrossEntropyLoss()(logits, m)
    dl = dice_loss(F.softmax(logits, dim=1).float(),
                   F.one_hot(m, n_classes).permute(0, 3, 1, 2).float(), multiclass=True)
    return ce, dl


def make_unet(name, n_classes, batch, h, w, seed, mask_mode="ellipse", bilinear=False):
    from unet import UNet
    from unet.evaluate import evaluate  # noqa: F401  (import check only)
    from util.dice_score import dice_loss, dice_coeff, multiclass_dice_coeff
    import torch.nn.functional as F

    sd = oracle.unet_state_dict(1, n_classes, seed=seed, bilinear=bilinear)
    net = UNet(1, n_classes, bilinear=bilinear)
    net.load_state_dict(sd, strict=True)
    net.train()
    size = max(h, w)
    x, mask = oracle.synthetic_batch(batch, size, seed=1234 + seed)
    x, mask = x[:, :, :h, :w].contiguous(), mask[:, :, :h, :w].contiguous()
    if mask_mode == "zeros":
        mask.zero_()
    elif mask_mode == "ones":
        mask.fill_(1)
    logits = net(x)
    l1, l2 = ref_seg_loss(logits, mask, dice_loss)
    loss = l1 + l2
    loss.backward()
    out = {
        "n_classes": n_classes, "seed": seed, "bilinear": int(bilinear), "x": x.numpy(), "mask": mask.numpy().astype(np.uint8),
        "logits": logits.detach().numpy(), "loss_ce": l1.item(), "loss_dice": l2.item(), "loss": loss.item(),
    }
    for k, p in net.named_parameters():
        out["gsum/" + k] = grad_summary(p.grad)
    for k, b in net.named_buffers():
        if "num_batches" not in k:
            out["buf/" + k] = b.detach().numpy().copy()
    for k, v in sd.items():
        if v.is_floating_point():
            out["wsum/" + k] = tensor_checksum(v)
    # eval-mode pass with the *updated* running statistics + the evaluate.py Dice
    net.eval()
    with torch.no_grad():
        lg = net(x)
        out["logits_eval"] = lg.numpy()
        if n_classes == 1:
            pred = (torch.sigmoid(lg) > 0.5).float()
            out["eval_dice"] = dice_coeff(pred.squeeze(), mask.float().squeeze(), reduce_batch_first=False).item()
        else:
            t = F.one_hot(mask[:, 0].long(), n_classes).permute(0, 3, 1, 2).float()
            p = F.one_hot(lg.argmax(dim=1), n_classes).permute(0, 3, 1, 2).float()
            out["eval_dice"] = multiclass_dice_coeff(p[:, 1:], t[:, 1:], reduce_batch_first=False).item()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "logits", tuple(logits.shape))


def _top_positions(t: torch.Tensor, k: int = 64) -> np.ndarray:
    """flat indices of the k largest-|t| elements (ties broken by index: deterministic)"""
    f = t.detach().abs().flatten()
    return torch.topk(f, k).indices.sort().values.numpy().astype(np.int64)


def make_unet_compact(name, n_classes, batch, size, seed):
    """BASELINE config 2 at its own size (UNet(1,2), 256x256, batch 32): a COMPACT fixture from the
    imported reference.  Inputs are not stored -- `oracle.synthetic_batch(batch, size, seed)` regenerates them (checksums
    stored); outputs: logits sub-sampled 8x8 plus the 64 largest-|logit| positions, the two losses, all gradient summaries,
    the updated BatchNorm buffers, eval-mode logits (same sub-sampling) and the evaluate.py Dice.  < 1 MB."""
    from unet import UNet
    from util.dice_score import dice_loss, dice_coeff, multiclass_dice_coeff
    import torch.nn.functional as F

    sd = oracle.unet_state_dict(1, n_classes, seed=seed)
    net = UNet(1, n_classes)
    net.load_state_dict(sd, strict=True)
    net.train()
    x, mask = oracle.synthetic_batch(batch, size, seed=1234 + seed)
    logits = net(x)
    l1, l2 = ref_seg_loss(logits, mask, dice_loss)
    loss = l1 + l2
    loss.backward()
    lg = logits.detach()
    top = _top_positions(lg)
    out = {
        "n_classes": n_classes, "seed": seed, "batch": batch, "size": size, "data_seed": 1234 + seed,
        "xsum": tensor_checksum(x), "masksum": tensor_checksum(mask.float()),
        "logits_sub": lg[:, :, ::8, ::8].numpy().copy(), "top_idx": top, "top_val": lg.flatten()[torch.from_numpy(top)].numpy(),
        # the FULL logits of the first four images (2 MB): the tolerance-meeting mode is checked over 0.5 M contiguous logits at
        # the size the metric is quoted on, not only on the 1.6 % sub-sample
        "logits_full4": lg[:4].numpy().copy(),
        "logits_absmean": lg.abs().mean().item(), "logits_sum": lg.double().sum().item(),
        "loss_ce": l1.item(), "loss_dice": l2.item(), "loss": loss.item(),
    }
    for k, p in net.named_parameters():
        out["gsum/" + k] = grad_summary(p.grad)
    for k, b in net.named_buffers():
        if "num_batches" not in k:
            out["buf/" + k] = b.detach().numpy().copy()
    for k, v in sd.items():
        if v.is_floating_point():
            out["wsum/" + k] = tensor_checksum(v)
    net.eval()
    with torch.no_grad():
        le = net(x)
        out["logits_eval_sub"] = le[:, :, ::8, ::8].numpy().copy()
        if n_classes == 1:
            pred = (torch.sigmoid(le) > 0.5).float()
            out["eval_dice"] = dice_coeff(pred.squeeze(), mask.float().squeeze(), reduce_batch_first=False).item()
        else:
            t = F.one_hot(mask[:, 0].long(), n_classes).permute(0, 3, 1, 2).float()
            p = F.one_hot(le.argmax(dim=1), n_classes).permute(0, 3, 1, 2).float()
            out["eval_dice"] = multiclass_dice_coeff(p[:, 1:], t[:, 1:], reduce_batch_first=False).item()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "logits", tuple(logits.shape))


def make_dice_cases():
    from util.dice_score import dice_coeff, dice_loss, multiclass_dice_coeff
    g = torch.Generator().manual_seed(7)
    out = {}
    p = torch.rand(3, 17, 13, generator=g)
    t = (torch.rand(3, 17, 13, generator=g) > 0.6).float()
    out["p"], out["t"] = p.numpy(), t.numpy()
    out["coeff_rbf_false"] = dice_coeff(p, t, False).item()
    out["coeff_rbf_true"] = dice_coeff(p, t, True).item()
    out["loss"] = dice_loss(p, t).item()
    z = torch.zeros_like(p)
    out["all_zero_loss"] = dice_loss(z, z).item()                 # dice == 1 -> loss 0
    out["zero_target_loss"] = dice_loss(p, z).item()
    out["coeff_2d"] = dice_coeff(p[0], t[0]).item()
    pm = torch.softmax(torch.randn(2, 3, 9, 11, generator=g), 1)
    tm = torch.nn.functional.one_hot(torch.randint(0, 3, (2, 9, 11), generator=g), 3).permute(0, 3, 1, 2).float()
    out["pm"], out["tm"] = pm.numpy(), tm.numpy()
    out["mc_rbf_false"] = multiclass_dice_coeff(pm, tm, False).item()
    out["mc_loss"] = dice_loss(pm, tm, multiclass=True).item()
    # gradient of the loss w.r.t. the probabilities
    pg = p.clone().requires_grad_(True)
    dice_loss(pg, t).backward()
    out["loss_grad_p"] = pg.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "dice_cases.npz"), **out)
    print("dice_cases done")


def make_jaccard_cases():
    """jaccard_index / jaccard_index_loss of running_files/train_end2end_isic.py:40-56.  The script itself cannot be
    imported (betty, wandb, imgaug, torchvision, CUDA at import time), so the two pure functions are compiled from its
    syntax tree here, in the build container, and only their inputs / outputs are stored."""
    import ast
    path = os.path.join(REF, "running_files", "train_end2end_isic.py")
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("jaccard_index", "jaccard_index_loss")]
    assert len(fns) == 2
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    jaccard_index, jaccard_index_loss = ns["jaccard_index"], ns["jaccard_index_loss"]
    g = torch.Generator().manual_seed(11)
    out = {}
    x = torch.randn(3, 1, 19, 14, generator=g) * 2
    t = (torch.rand(3, 1, 19, 14, generator=g) > 0.55).float()
    t[1] = 0                                                   # an empty mask
    out["logits"], out["mask"] = x.numpy(), t.numpy().astype(np.uint8)
    xg = x.clone().requires_grad_(True)
    bce = torch.nn.BCEWithLogitsLoss()(xg, t)                  # train_end2end_isic.py:248-249
    loss = bce + jaccard_index_loss(torch.sigmoid(xg.squeeze()), t.float().squeeze())
    loss.backward()
    out["loss"], out["bce"], out["grad"] = loss.item(), bce.item(), xg.grad.numpy()
    p = torch.sigmoid(x.squeeze())
    out["jaccard"] = jaccard_index(t.squeeze(), p).item()
    out["jaccard_2d"] = jaccard_index(t[0, 0], p[0]).item()   # the dim() == 2 branch (batch of one after squeeze)
    x1, t1 = x[:1], t[:1]
    out["loss_b1"] = (torch.nn.BCEWithLogitsLoss()(x1, t1) +
                      jaccard_index_loss(torch.sigmoid(x1.squeeze()), t1.float().squeeze())).item()
    np.savez_compressed(os.path.join(HERE, "jaccard_cases.npz"), **out)
    print("jaccard_cases done", out["loss"], out["jaccard"])


def make_ops_micro():
    """Per-op micro fixtures at tiny shapes from the reference's own modules."""
    from unet.unet_parts import DoubleConv, Down, Up, OutConv
    from models_pix2pix import networks
    # the modules below draw their default weights from the GLOBAL generator: seed it here, so that a full run (whose earlier
    # generators advance it) reproduces the same fixture as `--only ops` (hygiene flaw of this script)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(11)
    out = {}

    def rnd(*s):
        return torch.randn(*s, generator=g)

    # DoubleConv / Down / Up (odd sizes -> non-zero F.pad) / OutConv
    dc = DoubleConv(3, 8)
    x = rnd(2, 3, 9, 7)
    dc.train()
    y = dc(x)
    for k, v in dc.state_dict().items():
        out["dc/" + k] = v.numpy().copy()
    out["dc/x"], out["dc/y"] = x.numpy(), y.detach().numpy()
    dn = Down(4, 8)
    x = rnd(2, 4, 11, 9)
    for k, v in dn.state_dict().items():       # snapshot BEFORE the forward mutates BN buffers
        out["down/sd/" + k] = v.numpy().copy()
    out["down/x"], out["down/y"] = x.numpy(), dn(x).detach().numpy()
    upm = Up(8, 4, bilinear=False)
    x1, x2 = rnd(2, 8, 4, 3), rnd(2, 4, 9, 7)
    for k, v in upm.state_dict().items():
        out["up/sd/" + k] = v.numpy().copy()
    out["up/x1"], out["up/x2"], out["up/y"] = x1.numpy(), x2.numpy(), upm(x1, x2).detach().numpy()
    oc = OutConv(8, 2)
    x = rnd(2, 8, 5, 6)
    for k, v in oc.state_dict().items():
        out["outc/sd/" + k] = v.numpy().copy()
    out["outc/x"], out["outc/y"] = x.numpy(), oc(x).detach().numpy()

    # mixed transposed conv cell (networks.py:486-511) and its exact k=8 merge
    cell = networks.Cell_upconv(6, 5, bias=True, layer_index=2)
    arch = 0.3 * rnd(8, 3)
    networks.upconv_arch.data.copy_(arch)
    x = rnd(2, 6, 5, 4)
    for k, v in cell.state_dict().items():
        out["cell/sd/" + k] = v.numpy().copy()
    out["cell/arch"], out["cell/x"], out["cell/y"] = arch.numpy(), x.numpy(), cell(x).detach().numpy()

    # GANLoss modes
    pred = rnd(2, 1, 6, 6)
    out["gan/pred"] = pred.numpy()
    for mode in ("vanilla", "lsgan", "wgangp"):
        crit = networks.GANLoss(mode)
        out[f"gan/{mode}/real"] = crit(pred, True).item()
        out[f"gan/{mode}/fake"] = crit(pred, False).item()
    a, b = rnd(2, 1, 8, 8), rnd(2, 1, 8, 8)
    out["l1/a"], out["l1/b"], out["l1/y"] = a.numpy(), b.numpy(), torch.nn.L1Loss()(a, b).item()
    x, t = rnd(2, 1, 8, 8), (rnd(2, 1, 8, 8) > 0).float()
    out["bce/x"], out["bce/t"] = x.numpy(), t.numpy()
    out["bce/y"] = torch.nn.BCEWithLogitsLoss()(x, t).item()
    np.savez_compressed(os.path.join(HERE, "ops_micro.npz"), **out)
    print("ops_micro done")


def make_pix2pix(size=256, batch=2, name=None, compact=False):
    """compact=True (the batch-8 train-mode fixture): the real image is stored as uint8 levels (real = u8 / 255), the
    generator output sub-sampled 4x4, and the eval-mode outputs are left out -- a well-conditioned BatchNorm case that pins
    the generator's backward with the reference itself (batch 2 puts BN over 2..8 values at the deepest levels)."""
    from models_pix2pix import networks
    norm = networks.get_norm_layer("batch")
    G = networks.UnetGenerator(1, 1, 8, 64, norm_layer=norm, use_dropout=True)
    D = networks.NLayerDiscriminator(2, 64, 3, norm)
    sdG = seeded_generator_state_dict(seed=21)
    sdD = seeded_discriminator_state_dict(seed=22)
    G.load_state_dict(sdG, strict=True)
    D.load_state_dict(sdD, strict=True)
    g = torch.Generator().manual_seed(23)
    arch = 0.5 * torch.randn(8, 3, generator=g)
    networks.upconv_arch.data.copy_(arch)
    _, mask = oracle.synthetic_batch(batch, size, seed=77)
    real_mask = mask.float()
    real_image = torch.rand(batch, 1, size, size, generator=g)
    for m in G.modules():                     # dropout p forced to 0
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    out = {"arch": arch.numpy(), "mask": mask.numpy().astype(np.uint8)}
    if compact:
        u8 = (real_image * 256.0).floor().clamp(0, 255).to(torch.uint8)
        real_image = u8.float() / 255.0
        out["real_image_u8"] = u8.numpy()
    else:
        out["real_image"] = real_image.numpy()
    # train-mode generator + discriminator, the two GAN step losses and grads
    G.train(); D.train()
    fake = G(real_mask)
    out["fake_train"] = fake.detach().numpy()[:, :, ::4, ::4].copy() if compact else fake.detach().numpy()
    crit = networks.GANLoss("vanilla")
    pred_fake = D(torch.cat((real_mask, fake), 1))
    out["pred_fake_train"] = pred_fake.detach().numpy()
    loss_G = crit(pred_fake, True) + torch.nn.L1Loss()(fake, real_image) * 100.0
    G.zero_grad(); D.zero_grad()
    if networks.upconv_arch.grad is not None:
        networks.upconv_arch.grad = None
    loss_G.backward()
    out["loss_G"] = loss_G.item()
    for k, p in G.named_parameters():
        out["gsumG/" + k] = grad_summary(p.grad)
    out["arch_grad_G"] = networks.upconv_arch.grad.numpy().copy()
    for k, b in G.named_buffers():
        if "num_batches" not in k:
            out["bufG/" + k] = b.detach().numpy().copy()
    # D step (fresh modules so BN buffers start from the seeded state)
    D2 = networks.NLayerDiscriminator(2, 64, 3, norm)
    D2.load_state_dict(sdD, strict=True)
    D2.train()
    pf = D2(torch.cat((real_mask, fake), 1).detach())
    pr = D2(torch.cat((real_mask, real_image), 1))
    loss_D = (crit(pf, False) + crit(pr, True)) * 0.5
    loss_D.backward()
    out["loss_D"] = loss_D.item()
    out["pred_real_train"] = pr.detach().numpy()
    for k, p in D2.named_parameters():
        out["gsumD/" + k] = grad_summary(p.grad)
    if compact:
        for k, b in D2.named_buffers():
            if "num_batches" not in k:
                out["bufD/" + k] = b.detach().numpy().copy()
        np.savez_compressed(os.path.join(HERE, (name or f"pix2pix_{size}_b{batch}") + ".npz"), **out)
        print("pix2pix", size, "batch", batch, "loss_G", loss_G.item(), "loss_D", loss_D.item())
        return
    # eval-mode outputs from the seeded (un-stepped) state
    G3 = networks.UnetGenerator(1, 1, 8, 64, norm_layer=norm, use_dropout=True)
    G3.load_state_dict(sdG, strict=True)
    G3.eval()
    D3 = networks.NLayerDiscriminator(2, 64, 3, norm)
    D3.load_state_dict(sdD, strict=True)
    D3.eval()
    with torch.no_grad():
        fe = G3(real_mask)
        out["fake_eval"] = fe.numpy()
        out["pred_eval"] = D3(torch.cat((real_mask, fe), 1)).numpy()
    for k, v in sdD.items():
        if v.is_floating_point():
            out["wsumD/" + k] = tensor_checksum(v)
    for k, v in sdG.items():
        if v.is_floating_point():
            out["wsumG/" + k] = tensor_checksum(v)
    np.savez_compressed(os.path.join(HERE, f"pix2pix_{size}.npz"), **out)
    print("pix2pix", size, "loss_G", loss_G.item(), "loss_D", loss_D.item())


def make_unet3d(name, n_classes, batch, size, seed):
    """GenSeg-3D/UNet3D/unet3d.py imported with a stub `torchsummary` (its only use is the __main__ block, :8,134)."""
    import importlib.util, types
    sys.modules.setdefault("torchsummary", types.SimpleNamespace(summary=lambda *a, **k: None))
    spec = importlib.util.spec_from_file_location("ref_unet3d", os.path.join(REF, "GenSeg-3D", "UNet3D", "unet3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sd = oracle.unet3d_state_dict(1, n_classes, seed=seed)
    net = mod.UNet3D(in_channels=1, num_classes=n_classes)
    net.load_state_dict(sd, strict=True)
    net.train()
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(batch, 1, size, size, size, generator=g)
    mask = (torch.rand(batch, size, size, size, generator=g) > 0.6).long()
    logits = net(x)
    n, c, d, h, w = logits.shape
    loss = oracle.seg_loss(logits.reshape(n, c, d * h, w), mask.reshape(n, d * h, w))
    loss.backward()
    out = {"n_classes": n_classes, "seed": seed, "x": x.numpy(), "mask": mask.numpy().astype(np.uint8),
           "logits": logits.detach().numpy(), "loss": loss.item()}
    for k, p_ in net.named_parameters():
        out["gsum/" + k] = grad_summary(p_.grad)
    for k, b in net.named_buffers():
        out["buf/" + k] = b.detach().numpy().copy()
    for k, v in sd.items():
        if v.is_floating_point():
            out["wsum/" + k] = tensor_checksum(v)
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = net(x).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "logits", tuple(logits.shape))


def make_unet3d_compact(name, n_classes, size, seed):
    """BASELINE config 5's per-GPU size (UNet3D(1,2), one 128^3 volume): a compact fixture from the imported reference --
    inputs regenerated from the seed (checksums stored), logits sub-sampled 8x8x8 + the 64 largest-|logit| positions, loss,
    gradient summaries, BatchNorm buffers."""
    import importlib.util, types
    sys.modules.setdefault("torchsummary", types.SimpleNamespace(summary=lambda *a, **k: None))
    spec = importlib.util.spec_from_file_location("ref_unet3d", os.path.join(REF, "GenSeg-3D", "UNet3D", "unet3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sd = oracle.unet3d_state_dict(1, n_classes, seed=seed)
    net = mod.UNet3D(in_channels=1, num_classes=n_classes)
    net.load_state_dict(sd, strict=True)
    net.train()
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(1, 1, size, size, size, generator=g)
    mask = (torch.rand(1, size, size, size, generator=g) > 0.6).long()
    logits = net(x)
    n, c, d, h, w = logits.shape
    loss = oracle.seg_loss(logits.reshape(n, c, d * h, w), mask.reshape(n, d * h, w))
    loss.backward()
    lg = logits.detach()
    top = _top_positions(lg)
    out = {"n_classes": n_classes, "seed": seed, "size": size, "data_seed": 100 + seed,
           "xsum": tensor_checksum(x), "masksum": tensor_checksum(mask.float()),
           "logits_sub": lg[:, :, ::8, ::8, ::8].numpy().copy(), "top_idx": top,
           "top_val": lg.flatten()[torch.from_numpy(top)].numpy(), "logits_absmean": lg.abs().mean().item(),
           "loss": loss.item()}
    for k, p_ in net.named_parameters():
        out["gsum/" + k] = grad_summary(p_.grad)
    for k, b in net.named_buffers():
        out["buf/" + k] = b.detach().numpy().copy()
    for k, v in sd.items():
        if v.is_floating_point():
            out["wsum/" + k] = tensor_checksum(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "logits", tuple(logits.shape))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.only in ("", "unet"):
        make_unet("unet_c1_64", 1, 2, 64, 64, seed=1)
        make_unet("unet_c2_64", 2, 2, 64, 64, seed=2)
        make_unet("unet_c2_128_b4", 2, 4, 128, 128, seed=3)          # BASELINE config 1
        make_unet("unet_c1_odd", 1, 2, 70, 52, seed=4)               # non-zero F.pad through Up
        make_unet("unet_c1_zeros", 1, 2, 32, 32, seed=5, mask_mode="zeros")
        make_unet("unet_c1_ones", 1, 2, 32, 32, seed=6, mask_mode="ones")
    if a.only in ("", "unet", "bilinear"):
        make_unet("unet_c1_bilinear_64", 1, 2, 64, 64, seed=7, bilinear=True)      # nn.Upsample path, unet_parts.py:49-51
        make_unet("unet_c2_bilinear_odd", 2, 2, 70, 52, seed=8, bilinear=True)     # + non-zero F.pad
    if a.only in ("", "unet_b32"):
        make_unet_compact("unet_c2_256_b32", 2, 32, 256, seed=9)     # BASELINE config 2 at its own size (~2 min, ~30 GB)
    if a.only in ("", "ops"):
        make_dice_cases()
        make_ops_micro()
    if a.only in ("", "ops", "jaccard"):
        make_jaccard_cases()
    if a.only in ("", "pix2pix"):
        make_pix2pix(256, 2)
    if a.only in ("", "pix2pix", "pix2pix_b8"):
        make_pix2pix(256, 8, name="pix2pix_256_b8", compact=True)
    if a.only in ("", "unet3d"):
        make_unet3d("unet3d_c2_16", 2, 2, 16, seed=51)
        make_unet3d("unet3d_c1_16", 1, 2, 16, seed=52)
    if a.only in ("", "unet3d_128"):
        make_unet3d_compact("unet3d_c2_128", 2, 128, seed=53)        # BASELINE config 5, one volume per GPU
