"""mphip — MI355X-native Gbase hot slice for MegaPortraits.

`model` mirrors the reference's hot-path classes; `ops` are the per-kernel functional wrappers;
`_lib` is the ctypes binding of libmphip.so (include/mphip.h)."""
import os as _os



def request_hw_queues(n: int = 8) -> bool:
    """Several batches in flight = one C-side plan per caller stream, two HIP streams each.  The ROCm runtime multiplexes streams onto 4
    in-order hardware queues by default; with 8 every stream of two batches gets its own (same box: 3.65 -> 3.55 ms per step).  The
    variable GPU_MAX_HW_QUEUES is read when the HIP runtime starts and is process-wide, so the package does NOT touch it on import
    (ADVICE r4): an APPLICATION that wants several batches in flight calls this before its first HIP call (bench.py and the reenact
    CLI do).  An explicit setting of the variable wins.  Returns True when the request can still take effect."""
    import logging
    import sys

    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return True
    torch_mod = sys.modules.get("torch")
    started = bool(torch_mod is not None and torch_mod.cuda.is_initialized())
    if started:
        logging.warning("mphip.request_hw_queues(%d): the HIP runtime has already started, GPU_MAX_HW_QUEUES cannot take effect any more", n)
        return False
    _os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    logging.info("mphip: GPU_MAX_HW_QUEUES=%d requested for this process", n)
    return True


from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib", "ops", "model"]


def __getattr__(name):
    if name in ("ops", "model", "dp"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
