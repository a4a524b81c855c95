"""Import the reference `model.py` from /root/reference on CPU (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE — not part of the product path.  Only tests/, the golden
fixture generator (oracle/make_golden.py) and oracle validation may use it.

The reference cannot be imported as shipped: it pulls ~11 third-party packages
that are absent here (torchvision, cv2, lpips, mediapipe, ...) and
`mysixdrepnet.py:903` needs `numpy.lib.function_base` (removed in numpy 2).
None of those are touched by the hot-path classes (model.py:304-316, 369-471,
500-528, 571-597, 777-856, 927-1065), so they are replaced by inert stubs.

/root/reference does not exist on the GPU box: `reference_available()` is the
guard every caller must use; nothing under `-m gpu`, smoke() or bench.py calls
this module.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = {
    "torchvision", "cv2", "skimage", "lpips", "mediapipe", "facenet_pytorch",
    "face_recognition", "rembg", "colored_traceback", "torchsummary",
    "memory_profiler", "matplotlib", "PIL",
}


class _Stub(types.ModuleType):
    """Package-like module whose attributes are fabricated on demand."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []  # behave as a package so `import a.b.c` works

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        if item == "profile":  # memory_profiler.profile is used as a decorator
            return lambda f: f
        return mock.MagicMock(name=f"{self.__name__}.{item}")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        return None


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model.py"))


_cached = None


def load_reference_model():
    """Returns the imported reference `model` module (CPU)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not present (this only works in the build container)")
    present = set()
    for root in list(_STUB_ROOTS):
        try:
            __import__(root)
            present.add(root)
        except Exception:
            pass
    for root in present:
        _STUB_ROOTS.discard(root)
    sys.meta_path.insert(0, _StubFinder())
    if "numpy.lib.function_base" not in sys.modules:
        shim = types.ModuleType("numpy.lib.function_base")
        shim._quantile_unchecked = None
        sys.modules["numpy.lib.function_base"] = shim
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    saved = sys.modules.pop("model", None)
    try:
        import model as ref_model  # noqa: the reference's flat module
    finally:
        # do not leave the reference importable as `model` for anything else
        sys.modules.pop("model", None)
        if saved is not None:
            sys.modules["model"] = saved
        try:
            sys.path.remove(REFERENCE_ROOT)
        except ValueError:
            pass
    _cached = ref_model
    return ref_model
