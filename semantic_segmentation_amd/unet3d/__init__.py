from .unet3d import UNet3D, Conv3DBlock, UpConv3DBlock  # noqa: F401
