"""The four per-iteration training steps of running_files/train_end2end_jsrt.py (:141-236) as plain functions
over this package's modules -- the per-step math Betty's ImplicitProblems run, without Betty / wandb / imgaug.

    generator_step_loss      Generator.training_step      (:141-154)
    discriminator_step_loss  Discriminator.training_step  (:157-172)
    unet_step_loss           Unet.training_step           (:177-226)
    arch_step_loss           Arch.training_step           (:229-236)

Everything heavy runs on the HIP kernels (U-Net / generator / discriminator engines, fused seg loss, GAN / L1
reductions).  The fake-image post-processing of :197-200 (global min-max -> uint8 -> histogram equalise ->
gamma 0.5) is non-differentiable byte work on a detached tensor: three fused HIP launches (csrc/postproc.hip),
bit-exact against oracle/postproc.py's restatement of torchvision 0.14.1.  `mask_augment` stands in for the reference's host-side imgaug pipeline (:100-112, :186-190)."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from . import ops
from .losses import l1_loss, seg_loss


def generator_step_loss(netG, netD, criterionGAN, real_mask, real_image, lambda_L1: float = 100.0):
    fake_image = netG(real_mask)
    pred_fake = netD(torch.cat((real_mask, fake_image), 1))
    return criterionGAN(pred_fake, True) + l1_loss(fake_image, real_image) * lambda_L1


def discriminator_step_loss(netG, netD, criterionGAN, real_mask, real_image):
    with torch.no_grad():                         # the reference detaches the fake pair (:165)
        fake_image = netG(real_mask)
    pred_fake = netD(torch.cat((real_mask, fake_image), 1).detach())
    pred_real = netD(torch.cat((real_mask, real_image), 1))
    return (criterionGAN(pred_fake, False) + criterionGAN(pred_real, True)) * 0.5


_GAMMA_LUT = {}


def _gamma_lut(device, gamma: float) -> torch.Tensor:
    """The float an equalised uint8 level e ends up as: torchvision 0.14.1 `adjust_gamma` on a uint8 tensor
    (functional_tensor.py: convert_image_dtype to float32 = e/255, `** gamma` (sqrt for 0.5), clamp(0,1),
    convert_image_dtype back = mul(255 + 1 - 1e-3) truncated to uint8), followed by the script's `/ 255.0`
    (train_end2end_jsrt.py:200).  Built on the host in IEEE float32, one rounding per operation, then uploaded."""
    key = (str(device), gamma)
    if key not in _GAMMA_LUT:
        e = np.arange(256, dtype=np.float32) / np.float32(255.0)
        r = np.sqrt(e) if gamma == 0.5 else np.power(e, np.float32(gamma))
        r = np.clip(r.astype(np.float32), np.float32(0.0), np.float32(1.0))
        u8 = (r * np.float32(255.0 + 1.0 - 1e-3)).astype(np.uint8)
        _GAMMA_LUT[key] = torch.from_numpy(u8.astype(np.float32) / np.float32(255.0)).to(device).contiguous()
    return _GAMMA_LUT[key]


def fake_image_postprocess(fake_image: torch.Tensor, gamma: float = 0.5) -> torch.Tensor:
    """train_end2end_jsrt.py:197-200: global min-max -> uint8 -> per-(image, channel) histogram equalisation -> gamma
    -> float, as three HIP launches (gs_fake_postprocess).  Device tensors only: there is no host path."""
    if not fake_image.is_cuda or fake_image.dim() != 4:
        raise RuntimeError("fake_image_postprocess: needs a device tensor [N,C,H,W] (no CPU path)")
    x = fake_image.detach().float().contiguous()
    out = torch.empty_like(x)
    ops.fake_postprocess(x, out, _gamma_lut(x.device, gamma))
    return out


def isic_fake_trans(fake_image: torch.Tensor, generator: Optional[torch.Generator] = None, decisions=None) -> torch.Tensor:
    """The fake-image post-processing of the ISIC / RGB script (running_files/train_end2end_isic.py:178-184,263-264):
    global min-max -> uint8 -> fake_trans -> / 255, with fake_trans = Compose([RandomEqualize(p=0.5), RandomPosterize(4, p=1),
    RandomAdjustSharpness(0.3, p=0.5), RandomAutocontrast(p=0.5), ColorJitter(saturation=0.5)]) of torchvision 0.14.1.  As in
    torchvision every Random* transform draws ONE decision per call for the whole batch; they are drawn here on the host (in
    torchvision's order: four `rand(1) < p`, then ColorJitter's `randperm(4)` and `uniform_(0.5, 1.5)`) and the image work runs
    in six small HIP launches.  `decisions` = (equalize, sharpness, autocontrast, saturation_factor) overrides the draws."""
    if not fake_image.is_cuda or fake_image.dim() != 4:
        raise RuntimeError("isic_fake_trans: needs a device tensor [N,C,H,W] (no CPU path)")
    if decisions is None:
        r = lambda: float(torch.rand(1, generator=generator))
        eq_on = r() < 0.5
        r()                                           # RandomPosterize(p=1.0) draws too
        sharp_on = r() < 0.5
        ac_on = r() < 0.5
        torch.randperm(4, generator=generator)        # ColorJitter.get_params: order of its (single active) adjustment
        sat = float(torch.empty(1).uniform_(0.5, 1.5, generator=generator))
    else:
        eq_on, sharp_on, ac_on, sat = decisions
    x = fake_image.detach().float().contiguous()
    out = torch.empty_like(x)
    ops.isic_fake_trans(x, out, eq_on, 4, sharp_on, 0.3, ac_on, sat)
    return out


def unet_step_loss_isic(net, netG, images, true_masks, loss_lambda: float = 1.0,
                        mask_augment: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, generator=None):
    """Unet.training_step of running_files/train_end2end_isic.py:241-270: BCE + Jaccard loss on the real pair plus
    loss_lambda times the same loss on the generated RGB pair (G is cut by the detach)."""
    from .losses import seg_loss_jaccard
    loss = seg_loss_jaccard(net(images), true_masks)
    fake_mask = true_masks.float()
    if mask_augment is not None:
        fake_mask = mask_augment(fake_mask)
    fake_mask = torch.where(fake_mask > 0.1, torch.ones_like(fake_mask), torch.zeros_like(fake_mask))
    with torch.no_grad():
        fake_image = isic_fake_trans(netG(fake_mask), generator)
    return loss + loss_lambda * seg_loss_jaccard(net(fake_image), fake_mask)


def unet_step_losses(net, netG, images, true_masks, loss_lambda: float = 1.0,
                     mask_augment: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                     augmented_masks: Optional[torch.Tensor] = None):
    """The two terms of Unet.training_step (:176-226): (loss(net(images), masks), loss_lambda * loss(net(G(aug(masks))), aug(masks)));
    G is cut by the detach.  Their sum is the reference's loss; a caller that runs `.backward()` on each term in turn gets the same
    gradients (a + b in either order) with the second pass ACCUMULATING into `.grad` -- one foreach launch on this package's U-Net
    engine (engine.accumulate_grads) instead of autograd summing the two passes parameter by parameter.
    `augmented_masks`: aug(masks) computed by the caller (the HIP-graph form of the harness draws the augmentation on the host
    outside the captured step); `mask_augment` is not applied then."""
    loss = seg_loss(net(images), true_masks)
    fake_mask = true_masks.float() if augmented_masks is None else augmented_masks.float()
    if mask_augment is not None and augmented_masks is None:
        fake_mask = mask_augment(fake_mask)
    fake_mask = torch.where(fake_mask > 0.1, torch.ones_like(fake_mask), torch.zeros_like(fake_mask))
    with torch.no_grad():
        fake_image = fake_image_postprocess(netG(fake_mask))
    fake_loss = seg_loss(net(fake_image), fake_mask)
    return loss, loss_lambda * fake_loss


def unet_step_loss(net, netG, images, true_masks, loss_lambda: float = 1.0,
                   mask_augment: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                   augmented_masks: Optional[torch.Tensor] = None):
    """loss(net(images), masks) + loss_lambda * loss(net(G(aug(masks))), aug(masks)); G is cut by the detach."""
    loss, fake_loss = unet_step_losses(net, netG, images, true_masks, loss_lambda, mask_augment, augmented_masks)
    return loss + fake_loss

def arch_step_loss(net, image_valid, mask_valid):
    return seg_loss(net(image_valid), mask_valid)
