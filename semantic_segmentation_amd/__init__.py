"""MI355X-native (gfx950) implementation of the GenSeg segmentation hot path.

Public surface mirrors the reference's Python API for this path:
    from semantic_segmentation_amd.unet import UNet
    from semantic_segmentation_amd.util.dice_score import dice_loss, dice_coeff, multiclass_dice_coeff
    from semantic_segmentation_amd.losses import seg_loss
`semantic_segmentation_amd.compat.install()` additionally registers the reference's top-level module
names (`unet`, `util.dice_score`, ...) so the reference's run scripts import this package unchanged.
All arithmetic runs in libgsseg_hip.so (hand-written HIP); there is no CPU or ATen fallback."""
__version__ = "0.1.0"
