"""mphip — MI355X-native Gbase hot slice for MegaPortraits.

`model` mirrors the reference's hot-path classes; `ops` are the per-kernel functional wrappers;
`_lib` is the ctypes binding of libmphip.so (include/mphip.h)."""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib", "ops", "model"]


def __getattr__(name):
    if name in ("ops", "model", "dp"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
