from . import batch  # noqa: F401
