from . import pdipm  # noqa: F401
