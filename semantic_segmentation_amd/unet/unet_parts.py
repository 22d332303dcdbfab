"""U-Net building blocks with the reference's names, attributes and state-dict keys
(reference: unet/unet_parts.py:8-77).

The leaf modules (nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d) are PARAMETER CONTAINERS only: they
give identical parameter names, shapes, default initialisation and checkpoint compatibility.  Their ATen
forward is never used on the hot path -- `UNet.forward` runs the whole network through the hand-written
HIP kernels (unet_engine.py), which fuse across block boundaries (pool into the producer's BN pass, the
transposed conv straight into the concat buffer).  Calling a block on its own (`DoubleConv(x)`, `Down(x)`,
`Up(x1, x2)`, `OutConv(x)`, as the reference allows) runs the same kernels through block_engine.py: fp32 NCHW in
and out, first-order autograd, CPU tensors raise."""
import torch
import torch.nn as nn


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   (unet_parts.py:8-25)"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        from .block_engine import double_conv_forward
        return double_conv_forward(self, x)


class Down(nn.Module):
    """Downscaling with maxpool then double conv   (unet_parts.py:28-39)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        from .block_engine import down_forward
        return down_forward(self, x)


class Up(nn.Module):
    """Upscaling then double conv   (unet_parts.py:42-68)"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        from .block_engine import up_forward
        return up_forward(self, x1, x2)


class OutConv(nn.Module):
    """1x1 convolution head   (unet_parts.py:71-77)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        from .block_engine import out_conv_forward
        return out_conv_forward(self, x)
