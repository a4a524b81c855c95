"""mphip — MI355X-native Gbase hot slice for MegaPortraits.

`model` mirrors the reference's hot-path classes; `ops` are the per-kernel functional wrappers;
`_lib` is the ctypes binding of libmphip.so (include/mphip.h)."""
import os as _os



def request_hw_queues(n: int = 8) -> bool:
    """Opt-in, for a process whose ONLY GPU work is the hot slice with several batches in flight (one C-side plan per caller stream, two
    HIP streams each).  The ROCm runtime multiplexes streams onto 4 in-order hardware queues by default; with 8 every stream of two
    batches gets its own (same box, r06: 3026 -> 3068 frames/s, +1.3 %).  NOT for a process that also runs other work between hot-slice
    calls (gbase.Gbase.forward, the reenact CLI, a training loop): with more queues than the device keeps resident a queue that goes idle
    is re-scheduled with a delay on EVERY dispatch, and a plan whose side stream sits on such a queue turns its 36 dependent generator
    launches into ~28 ms per call (r05's `end_to_end_autocast_fp16` 118 -> 71 frames/s; profiles/NOTES_r06.md §1).  Nothing in this
    repository calls it any more (bench.py: `--hw-queues 8`).  GPU_MAX_HW_QUEUES is read when the HIP runtime starts and is process-wide;
    an explicit setting of the variable wins.  Returns True when the request can still take effect."""
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
