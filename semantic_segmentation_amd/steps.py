"""The four per-iteration training steps of running_files/train_end2end_jsrt.py (:141-236) as plain functions
over this package's modules -- the per-step math Betty's ImplicitProblems run, without Betty / wandb / imgaug.

    generator_step_loss      Generator.training_step      (:141-154)
    discriminator_step_loss  Discriminator.training_step  (:157-172)
    unet_step_loss           Unet.training_step           (:177-226)
    arch_step_loss           Arch.training_step           (:229-236)

Everything heavy runs on the HIP kernels (U-Net / generator / discriminator engines, fused seg loss, GAN / L1
reductions).  The fake-image post-processing of :197-200 (global min-max -> uint8 -> histogram equalise ->
gamma 0.5) is non-differentiable, operates on a detached tensor and is done with a few device-side torch ops
here; SURVEY section 8(f) ranks a fused kernel for it (with the on-device mask augmentation) as the next
component.  `mask_augment` stands in for the reference's host-side imgaug pipeline (:100-112, :186-190)."""
from __future__ import annotations

from typing import Callable, Optional

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


def equalize_gamma_u8(img_u8: torch.Tensor, gamma: float = 0.5) -> torch.Tensor:
    """torchvision.transforms.functional.equalize + adjust_gamma on uint8 [N,1,H,W] (per image), device side.
    equalize: lut = (cumsum(hist) - hist + step//2) // step with step = (sum(hist) - last nonzero hist) // 255;
    identity when step == 0.  adjust_gamma on uint8: (255 * (v/255)^gamma) truncated to uint8."""
    n = img_u8.shape[0]
    flat = img_u8.reshape(n, -1).long()
    hist = torch.zeros((n, 256), dtype=torch.long, device=img_u8.device)
    hist.scatter_add_(1, flat, torch.ones_like(flat))
    idx = torch.arange(256, device=img_u8.device).expand(n, 256)
    last_nz = torch.where(hist > 0, idx, torch.full_like(idx, -1)).max(dim=1).values
    last_val = hist.gather(1, last_nz.clamp_min(0).unsqueeze(1)).squeeze(1)
    step = torch.div(hist.sum(1) - last_val, 255, rounding_mode="floor")
    csum = torch.cumsum(hist, 1) - hist                       # exclusive prefix sum
    safe = step.clamp_min(1).unsqueeze(1)
    lut = torch.div(csum + torch.div(safe, 2, rounding_mode="floor"), safe, rounding_mode="floor").clamp(0, 255)
    lut = torch.where((step == 0).unsqueeze(1), idx, lut)
    eq = lut.gather(1, flat).reshape(img_u8.shape)
    out = (255.0 * (eq.float() / 255.0) ** gamma).clamp(0, 255).to(torch.uint8)
    return out


_GAMMA_LUT = {}


def _gamma_lut(device, gamma: float) -> torch.Tensor:
    """float value the pipeline maps an equalised uint8 level to (same torch expression as equalize_gamma_u8, on the device)."""
    key = (str(device), gamma)
    if key not in _GAMMA_LUT:
        e = torch.arange(256, device=device, dtype=torch.float32)
        _GAMMA_LUT[key] = ((255.0 * (e / 255.0) ** gamma).clamp(0, 255).to(torch.uint8).float() / 255.0).contiguous()
    return _GAMMA_LUT[key]


def fake_image_postprocess(fake_image: torch.Tensor, fused: bool = True) -> torch.Tensor:
    """train_end2end_jsrt.py:197-200.  One-channel batches run as three fused HIP launches (gs_fake_postprocess), bit-identical
    to the torch expression below (`fused=False`)."""
    if fused and fake_image.is_cuda and fake_image.dim() == 4 and fake_image.shape[1] == 1:
        x = fake_image.detach().float().contiguous()
        out = torch.empty_like(x)
        ops.fake_postprocess(x, out, _gamma_lut(x.device, 0.5))
        return out
    f = ((fake_image - fake_image.min()) / (fake_image.max() - fake_image.min())).detach()
    u8 = f.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)
    return equalize_gamma_u8(u8, 0.5).float() / 255.0


def unet_step_loss(net, netG, images, true_masks, loss_lambda: float = 1.0,
                   mask_augment: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
    """loss(net(images), masks) + loss_lambda * loss(net(G(aug(masks))), aug(masks)); G is cut by the detach."""
    loss = seg_loss(net(images), true_masks)
    fake_mask = true_masks.float()
    if mask_augment is not None:
        fake_mask = mask_augment(fake_mask)
    fake_mask = torch.where(fake_mask > 0.1, torch.ones_like(fake_mask), torch.zeros_like(fake_mask))
    with torch.no_grad():
        fake_image = fake_image_postprocess(netG(fake_mask))
    fake_loss = seg_loss(net(fake_image), fake_mask)
    return loss + loss_lambda * fake_loss


def arch_step_loss(net, image_valid, mask_valid):
    return seg_loss(net(image_valid), mask_valid)
