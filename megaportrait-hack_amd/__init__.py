"""mphip — MI355X-native Gbase hot slice for MegaPortraits.

`model` mirrors the reference's hot-path classes; `ops` are the per-kernel functional wrappers;
`_lib` is the ctypes binding of libmphip.so (include/mphip.h)."""
import os as _os

# Several batches in flight = one C-side plan per caller stream, two HIP streams each.  The ROCm runtime multiplexes streams onto 4
# in-order hardware queues by default; with 8 every stream of two batches gets its own (same box: 3.65 -> 3.55 ms per step,
# DESIGN.md 3 "Late r03").  The variable is read when the HIP runtime starts, so it is set at import — an explicit setting wins, and
# model._plan_for warns when a second plan appears without it.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib", "ops", "model"]


def __getattr__(name):
    if name in ("ops", "model", "dp"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
