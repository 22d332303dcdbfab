"""Registers the reference's top-level module names so its run scripts import this package unchanged
(`from unet import UNet`, `from unet.evaluate import evaluate`, `from util.dice_score import dice_loss`,
`from models_pix2pix import create_model, networks` -- running_files/train_end2end_jsrt.py:22-31).

Only the hot-path modules are taken over.  Everything else of the caller's tree stays its own Python: in
particular the caller's `util` package keeps resolving (`from util import util`, `from util.JSRT_loader import ...`,
train_end2end_jsrt.py:22-24; options/base_options.py:3) and only its `dice_score` submodule is replaced; `options.*`
is never touched (it reaches this package through `models_pix2pix.get_option_setter`, base_options.py:5,65-71)."""
import importlib
import importlib.util
import os
import sys
import types

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def _is_ours(mod) -> bool:
    f = getattr(mod, "__file__", None) or ""
    return os.path.abspath(f).startswith(_PKG_DIR + os.sep)


def _callers_package(name):
    """The `name` package of the caller's own tree (already imported, or importable from sys.path), or None."""
    mod = sys.modules.get(name)
    if mod is not None and not _is_ours(mod) and getattr(mod, "__path__", None):
        return mod
    if mod is not None and not _is_ours(mod) and not getattr(mod, "__is_gsseg_stub__", False):
        return None                        # the caller's own single-file module of that name: never evict it
    if mod is not None:
        del sys.modules[name]              # an earlier stub of ours / our namesake: look at sys.path again
    try:
        spec = importlib.util.find_spec(name)
    except (ImportError, ValueError):
        spec = None
    if spec is None or spec.submodule_search_locations is None:
        return None
    return importlib.import_module(name)


def install(force: bool = True) -> None:
    """force=True (default): the hot-path names are (re)bound even if the caller's own modules of that name were
    imported before; force=False keeps whatever is already in sys.modules.  The caller's `util` package is never
    shadowed in either mode."""
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

    util = _callers_package("util")
    foreign_file = sys.modules.get("util")
    if util is None and foreign_file is not None and not _is_ours(foreign_file):
        # the caller has a single-file `util.py`: it stays what `import util` gives; only `util.dice_score` is added
        # (as a sys.modules entry and as an attribute), `from util import X` keeps working
        util = foreign_file
    elif util is None:
        # no `util` package anywhere on sys.path: provide a bare one so `import util.dice_score` resolves
        util = types.ModuleType("util")
        util.__path__ = []
        util.__is_gsseg_stub__ = True
        sys.modules["util"] = util
    if force or "util.dice_score" not in sys.modules:
        sys.modules["util.dice_score"] = _dice
        util.dice_score = _dice

    # Pix2Pix side: `from models_pix2pix import create_model, networks` (train_end2end_jsrt.py:28-29,55-59) and the
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
