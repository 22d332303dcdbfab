"""Full assembly of the U-Net (reference: unet/unet_model.py:8-49) on the MI355X HIP engine.

Drop-in surface: `UNet(n_channels, n_classes, bilinear=False)`, attributes `.n_channels .n_classes
.bilinear`, children `inc down1..4 up1..4 outc`, 118 state-dict keys identical to the reference, fp32
NCHW in -> fp32 NCHW logits out, first-order autograd.  `forward` always runs the hand-written kernels;
there is no ATen / CPU fallback (a CPU tensor or a missing libgsseg_hip.so raises)."""
import os

import torch.nn as nn

from .unet_engine import UNetEngine, run_unet
from .unet_parts import DoubleConv, Down, OutConv, Up


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=False, compute_dtype=None, precise=None, dynamic_loss_scale=None):
        super(UNet, self).__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear

        self.inc = (DoubleConv(n_channels, 64))
        self.down1 = (Down(64, 128))
        self.down2 = (Down(128, 256))
        self.down3 = (Down(256, 512))
        factor = 2 if bilinear else 1
        self.down4 = (Down(512, 1024 // factor))
        self.up1 = (Up(1024, 512 // factor, bilinear))
        self.up2 = (Up(512, 256 // factor, bilinear))
        self.up3 = (Up(256, 128 // factor, bilinear))
        self.up4 = (Up(128, 64, bilinear))
        self.outc = (OutConv(64, n_classes))
        # 16-bit storage/MFMA dtype of the engine: fp16 (default: 8x finer mantissa than bf16 at the same
        # MFMA rate; gradients are loss-scaled internally) or bf16.  Not part of the state dict.
        dt = compute_dtype or os.environ.get("GSSEG_DTYPE", "f16")
        # Numerics mode of the forward.  The reference's forward is plain fp32 (unet/unet_model.py:26-37) and the
        # north star asks for logits within 1e-3 of it, so the DEFAULT is the mode that meets that bound:
        #   precise=None / "auto" (GSSEG_PRECISE unset or "auto"): the "mixed" pair forward -- every tensor a hi/lo pair of 16-bit
        #       values, correction MFMA segments on the stages that make the 16-bit error (unet_engine.MIXED_XW): max |dlogit| < 1e-3.
        #       Configurations the pair forward does not cover (more than 4 input channels / classes) run the 16-bit engine.
        #   precise="mixed" (GSSEG_PRECISE=mixed): the same, but unsupported configurations raise instead of falling back.
        #   precise=True (GSSEG_PRECISE=1): three MFMA segments everywhere, logits ~1e-5 from fp32 at ~3x the forward MFMA work.
        #   precise=False (GSSEG_PRECISE=0): the FAST mode -- single 16-bit storage, logits within ~4e-3, loss / Dice within 1e-5.
        if precise is None:
            env = os.environ.get("GSSEG_PRECISE", "auto")
            precise = {"0": False, "1": True, "": "auto"}.get(env, env)
        object.__setattr__(self, "_engine", UNetEngine(self, dt, precise=precise))
        # fp16 backward: gradients are carried times a static power-of-two scale that assumes a mean-reduced loss;
        # dynamic_loss_scale=True renormalises the incoming gradient on the device (sum-reduced losses, GradScaler, ...)
        if dynamic_loss_scale is not None:
            self._engine.dynamic_loss_scale = bool(dynamic_loss_scale)

    @property
    def engine(self) -> UNetEngine:
        return self._engine

    def forward(self, x):
        return run_unet(self._engine, x)

    def use_checkpointing(self):
        """Kept for API parity (unet_model.py:39-49; broken upstream: it calls the *module*
        torch.utils.checkpoint).  Activation recomputation is not needed at 288 GB of HBM: no-op."""
        return None
