"""models_pix2pix package surface of the reference (models_pix2pix/__init__.py:22-67): `create_model(opt)`,
`get_option_setter(name)`, `networks`.  Only `--model pix2pix` is on the MI355X hot path."""
from . import networks  # noqa: F401


def find_model_using_name(model_name):
    if model_name.replace("_", "").lower() != "pix2pix":
        raise NotImplementedError("model [%s] is not on the MI355X hot path (only pix2pix)" % model_name)
    from .pix2pix_model import Pix2PixModel
    return Pix2PixModel


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(getattr(opt, "model", "pix2pix"))(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
