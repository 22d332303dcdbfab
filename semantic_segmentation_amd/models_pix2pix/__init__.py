from . import networks  # noqa: F401
