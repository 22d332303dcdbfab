"""Registers the reference's top-level module names so its run scripts import this package unchanged
(`from unet import UNet`, `from unet.evaluate import evaluate`, `from util.dice_score import dice_loss`).
Only the hot-path modules are provided; everything else of the reference stays its own Python."""
import sys
import types


def install(force: bool = True) -> None:
    from . import unet as _unet
    from .unet import evaluate as _evaluate, unet_model as _unet_model, unet_parts as _unet_parts
    from .util import dice_score as _dice

    def put(name, mod):
        if force or name not in sys.modules:
            sys.modules[name] = mod

    put("unet", _unet)
    put("unet.unet_model", _unet_model)
    put("unet.unet_parts", _unet_parts)
    put("unet.evaluate", _evaluate)
    util = sys.modules.get("util")
    if util is None or force:
        util = types.ModuleType("util")
        util.__path__ = []          # namespace-like: lets `import util.dice_score` resolve from sys.modules
        put("util", util)
    util.dice_score = _dice
    put("util.dice_score", _dice)
    # Pix2Pix side: `from models_pix2pix import create_model, networks` (train_end2end_jsrt.py:17,55-59) and the
    # architecture package networks.py imports (networks.py:8-9)
    from . import architecture_pix2pix as _arch, models_pix2pix as _mp
    from .architecture_pix2pix import genotypes as _geno, operations as _opsmod
    from .models_pix2pix import networks as _networks, pix2pix_model as _pm
    put("models_pix2pix", _mp)
    put("models_pix2pix.networks", _networks)
    put("models_pix2pix.pix2pix_model", _pm)
    put("architecture_pix2pix", _arch)
    put("architecture_pix2pix.genotypes", _geno)
    put("architecture_pix2pix.operations", _opsmod)
