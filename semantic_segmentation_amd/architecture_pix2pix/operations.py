"""Transposed-conv primitives of the mixed up-conv cell (reference: architecture_pix2pix/operations.py:14-39).
Inside UnetGenerator.forward the three primitives of a cell run merged into one 8x8 kernel on the HIP engine; a primitive
called on its own (`re_conv_421(C_in, C_out, bias)(x)`, as the reference allows) is the one-hot case of the same kernels
(models_pix2pix/cell_engine.py): fp32 NCHW in and out, first-order autograd, CPU tensors raise."""
import torch.nn as nn

UPCONV_KSP = {'re_conv_421': (4, 2, 1), 're_conv_622': (6, 2, 2), 're_conv_823': (8, 2, 3)}


class _ReConv(nn.Module):
    KSP = (4, 2, 1)

    def __init__(self, C_in, C_out, bias):
        super().__init__()
        k, s, p = self.KSP
        self.op = nn.ConvTranspose2d(C_in, C_out, kernel_size=k, stride=s, padding=p, bias=bias)

    def forward(self, x):
        from ..models_pix2pix.cell_engine import single_upconv
        return single_upconv(self, x)


class re_conv_421(_ReConv):
    KSP = (4, 2, 1)


class re_conv_622(_ReConv):
    KSP = (6, 2, 2)


class re_conv_823(_ReConv):
    KSP = (8, 2, 3)


OPS = {
    're_conv_421': lambda C_in, C_out, bias: re_conv_421(C_in, C_out, bias),
    're_conv_622': lambda C_in, C_out, bias: re_conv_622(C_in, C_out, bias),
    're_conv_823': lambda C_in, C_out, bias: re_conv_823(C_in, C_out, bias),
}
