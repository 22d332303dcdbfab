from .unet_model import UNet  # noqa: F401
