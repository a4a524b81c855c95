"""Functional wrappers: torch CUDA tensors in/out, HIP kernels (libmphip.so, C ABI) in between.

PyTorch is used here only for device memory (tensor allocation through its caching allocator,
which is stream-ordered) and for the current HIP stream.  Every function requires float32,
contiguous CUDA tensors and raises otherwise — there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import os as _os

import torch

from . import _lib

_P = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else _P(t.data_ptr())


def _stream():
    return _P(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA tensor (the HIP path has no CPU fallback), got "
                           f"{type(t).__name__} on {getattr(t, 'device', None)}")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # kernels launch on the calling thread's current device and stream (one process per GPU)
        raise RuntimeError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(tensor.device) (or use `with torch.cuda.device(...)`) first")
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------ range descriptors (f16x3 operand scales)
# include/mphip.h "Range descriptors": a small device buffer per tensor; the f16x3 conv kernels scale their input by the
# power of two it implies, so activations of any magnitude keep fp32-class accuracy and nothing is ever clamped.  Kernels
# that already stream a tensor (warp gather, GroupNorm apply) fill the descriptor of their output for free; it rides on the
# torch.Tensor object as an attribute, guarded by the tensor's version counter (an in-place write invalidates it).
_RANGE_FLOATS = 4100  # MPHIP_RANGE_FLOATS (include/mphip.h): 4 + one partial maximum per producing workgroup
_RANGES_ENABLED = _os.environ.get("MPHIP_FUSED_RANGES", "1") != "0"  # dev switch: 0 = every f16x3 conv measures its own input


_POISON_RANGES = _os.environ.get("MPHIP_POISON_RANGES", "0") == "1"  # tests: a slot nobody wrote shows up as a 1e30 maximum


def new_range(device) -> torch.Tensor:
    if _POISON_RANGES:
        return torch.full((_RANGE_FLOATS,), 1.0e30, dtype=torch.float32, device=device)
    return torch.empty(_RANGE_FLOATS, dtype=torch.float32, device=device)  # filled by the kernel call it is handed to


def tag_range(t: torch.Tensor, rng: Optional[torch.Tensor]) -> torch.Tensor:
    if rng is not None:
        t._mphip_range = (rng, t._version)
        # produced by a kernel that is itself part of the capture in progress (replays refill it): remember WHICH capture
        t._mphip_range_capture = _capture_epoch if torch.cuda.is_current_stream_capturing() else -1
    return t


# Module-global state: `_capture_epoch`, `_repack_always` / `_repack_token`, `_weight_epoch`, `_conv_hook`, `_default_precision`
# are PROCESS-wide on purpose — a capture's forward runs on the caller's thread and its backward on autograd's device thread, and both
# must see the same "this is a capture" state (a thread-local would hide it from the backward's pack lookups).  The counters are bumped
# under a lock; the flags are configuration of the process: one capture at a time (torch.cuda.graph is not re-entrant either), hooks and
# the default precision set before the threads that launch are started.
import threading as _threading

_state_lock = _threading.Lock()
_capture_epoch = 0


def begin_capture() -> None:
    """Called by GraphedHotSlice / training.GraphedTrainStep right before `torch.cuda.graph(...)`: descriptors noted during an
    earlier capture (or during warm-up) are not trusted inside the new one."""
    global _capture_epoch
    with _state_lock:
        _capture_epoch += 1


def tensor_range(t: torch.Tensor) -> Optional[torch.Tensor]:
    hit = getattr(t, "_mphip_range", None)
    return hit[0] if (hit is not None and hit[1] == t._version) else None


def absmax_range(x: torch.Tensor) -> torch.Tensor:
    """Range descriptor of a tensor of unknown origin: one streaming pass (mphip_absmax_range); cached on the tensor."""
    x = _req(x, "x")
    rng = new_range(x.device)
    _lib.check(_lib.load().mphip_absmax_range(_ptr(x), x.numel(), _ptr(rng), _stream()), "mphip_absmax_range")
    tag_range(x, rng)
    return rng


def _range_for(x: torch.Tensor, given: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Operand-scale descriptor of a conv input: the one its producer noted, else one streaming pass.
    While a hipGraph is being captured a descriptor cached on a tensor of unknown origin is NOT
    trusted: it was measured on the warm-up batch, and a cache hit would record no absmax kernel — every replay would
    then scale new inputs by the warm-up batch's range (overflow to Inf for larger inputs, lost precision for smaller
    ones).  Descriptors produced INSIDE the capture (warp gather, GroupNorm apply) are recorded with it and stay valid."""
    if given is not None:
        return given
    hit = tensor_range(x)
    if hit is not None and torch.cuda.is_current_stream_capturing() and getattr(x, "_mphip_range_capture", -1) != _capture_epoch:
        hit = None
    return hit if hit is not None else absmax_range(x)


# ------------------------------------------------------------------ host-built tables
_tables = {}
_captured = None


def _captured_tables():
    """fp32 bit patterns of the reference host's torch.linspace tables (data/linspace_tables.json,
    written by oracle/make_golden.py:capture_tables).  ATen's linspace has no closed form and its
    bits depend on the host ISA; the hot-path sizes (16, 64) are therefore
    data, which keeps the index pipeline identical on every host.  Other sizes fall back to the
    local host's torch.linspace."""
    global _captured
    if _captured is None:
        import json
        import os

        import numpy as np

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "linspace_tables.json")
        with open(path) as f:
            raw = json.load(f)
        _captured = {kind: {int(n): torch.from_numpy(np.array(bits, dtype=np.uint32).view(np.float32).copy())
                            for n, bits in raw[kind].items()} for kind in ("linspace", "affine_base")}
    return _captured


def linspace_table(n: int, device) -> torch.Tensor:
    """torch.linspace(-1,1,n) as the reference's CPU path evaluates it (model.py:1040-1042)."""
    key = ("lin", n, str(device))
    if key not in _tables:
        cap = _captured_tables()["linspace"].get(n)
        host = cap if cap is not None else torch.linspace(-1, 1, n, dtype=torch.float32)
        _tables[key] = host.to(device)
    return _tables[key]


def affine_base_table(g: int, device) -> torch.Tensor:
    """Base coordinates of F.affine_grid(..., align_corners=False): linspace(-1,1,G)*(G-1)/G."""
    key = ("aff", g, str(device))
    if key not in _tables:
        cap = _captured_tables()["affine_base"].get(g)
        host = cap if cap is not None else torch.linspace(-1, 1, g, dtype=torch.float32) * (g - 1) / g
        _tables[key] = host.to(device)
    return _tables[key]


# ------------------------------------------------------------------ K0 / K1
def rt_theta(rotation_deg: torch.Tensor, translation: torch.Tensor, invert: bool) -> torch.Tensor:
    rotation_deg = _req(rotation_deg, "rotation")
    translation = _req(translation, "translation")
    b = rotation_deg.shape[0]
    if rotation_deg.shape != (b, 3) or translation.shape != (b, 3):
        raise RuntimeError(f"rt_theta: expected [B,3] and [B,3], got {tuple(rotation_deg.shape)} {tuple(translation.shape)}")
    theta = torch.empty((b, 3, 4), dtype=torch.float32, device=rotation_deg.device)
    lib = _lib.load()
    _lib.check(lib.mphip_rt_theta(_ptr(rotation_deg), _ptr(translation), _ptr(theta), b, int(bool(invert)), _stream()),
               "mphip_rt_theta")
    return theta


def warp_field_compose(theta: torch.Tensor, em: torch.Tensor, grid_size: int = 64, parts: bool = False):
    theta = _req(theta, "theta")
    em = _req(em, "em")
    b = theta.shape[0]
    if theta.shape != (b, 3, 4) or em.dim() != 5 or em.shape[0] != b or em.shape[1] != 3:
        raise RuntimeError(f"warp_field_compose: bad shapes theta={tuple(theta.shape)} em={tuple(em.shape)}")
    g = grid_size
    w = torch.empty((b, 3, g, g, g), dtype=torch.float32, device=theta.device)
    rt = torch.empty_like(w) if parts else None
    e64 = torch.empty_like(w) if parts else None
    lib = _lib.load()
    _lib.check(lib.mphip_warp_field_compose(_ptr(theta), _ptr(em), _ptr(affine_base_table(g, theta.device)), _ptr(w),
                                            _ptr(rt), _ptr(e64), b, em.shape[2], em.shape[3], em.shape[4], g, _stream()),
               "mphip_warp_field_compose")
    return (w, rt, e64) if parts else w


# ------------------------------------------------------------------ K2 / K3
def warp_volume(v: torch.Tensor, field: torch.Tensor, return_coords: bool = False):
    v = _req(v, "v")
    field = _req(field, "warp_field")
    if v.dim() != 5 or field.dim() != 5 or field.shape[0] != v.shape[0] or field.shape[1] != 3:
        raise RuntimeError(f"warp_volume: bad shapes v={tuple(v.shape)} field={tuple(field.shape)}")
    b, c, d, h, w = v.shape
    out = torch.empty_like(v)
    rng = new_range(v.device) if _RANGES_ENABLED else None
    coords = idx = None
    if return_coords:
        coords = torch.empty((b, d, h, w, 3), dtype=torch.float32, device=v.device)
        idx = torch.empty((b, d, h, w, 3), dtype=torch.int32, device=v.device)
    dev = v.device
    lib = _lib.load()
    ws_bytes = lib.mphip_warp_workspace_bytes(b, d, h, w) + lib.mphip_warp_corner_image_bytes(b, c)   # (+ K2's optional corner image)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    _lib.check(lib.mphip_warp_volume(_ptr(v), _ptr(field), _ptr(linspace_table(d, dev)), _ptr(linspace_table(h, dev)),
                                     _ptr(linspace_table(w, dev)), _ptr(out), _ptr(coords), _ptr(idx), _ptr(rng), b, c, d, h, w,
                                     field.shape[2], field.shape[3], field.shape[4], _ptr(ws), ws_bytes, _stream()),
               "mphip_warp_volume")
    tag_range(out, rng)
    return (out, coords, idx) if return_coords else out


def warp_volume_dsum(v: torch.Tensor, field: torch.Tensor) -> torch.Tensor:
    v = _req(v, "v")
    field = _req(field, "warp_field")
    if v.dim() != 5 or field.dim() != 5 or (field.shape[0] != v.shape[0] and v.shape[0] != 1) or field.shape[1] != 3:
        raise RuntimeError(f"warp_volume_dsum: bad shapes v={tuple(v.shape)} field={tuple(field.shape)}")
    _, c, d, h, w = v.shape
    b = field.shape[0]
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=v.device)
    dev = v.device
    lib = _lib.load()
    ws_bytes = lib.mphip_warp_workspace_bytes(b, d, h, w)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    # one source volume for B driver fields (dp.cross_reenact): the kernel reads it in place, no expanded copy
    fn = lib.mphip_warp_volume_dsum_shared if (v.shape[0] == 1 and b > 1) else lib.mphip_warp_volume_dsum
    _lib.check(fn(_ptr(v), _ptr(field), _ptr(linspace_table(d, dev)), _ptr(linspace_table(h, dev)),
                                          _ptr(linspace_table(w, dev)), _ptr(out), b, c, d, h, w, field.shape[2],
                                          field.shape[3], field.shape[4], _ptr(ws), ws_bytes, _stream()),
               "mphip_warp_volume_dsum")
    return out


def warp_coords(field: torch.Tensor, d: int, h: int, w: int) -> torch.Tensor:
    """K2/K3's coordinate pass alone: field [B,3,fD,fH,fW] -> clipped sample positions [B,d,h,w,3] (x,y,z)."""
    field = _req(field, "warp_field")
    b = field.shape[0]
    dev = field.device
    coords = torch.empty((b, d, h, w, 3), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().mphip_warp_coords(_ptr(field), _ptr(linspace_table(d, dev)), _ptr(linspace_table(h, dev)), _ptr(linspace_table(w, dev)),
                                             _ptr(coords), b, d, h, w, field.shape[2], field.shape[3], field.shape[4], _stream()), "mphip_warp_coords")
    return coords


def warp_sample_box(coords: torch.Tensor) -> torch.Tensor:
    """Per frame the box of source voxels the samples touch: int32 [B,8] = {lx, ly, lz, ex, ey, ez, -, -} (include/mphip.h
    "demand-driven evaluation")."""
    coords = _req(coords, "coords")
    b, d, h, w, _ = coords.shape
    box = torch.empty((b, 8), dtype=torch.int32, device=coords.device)
    _lib.check(_lib.load().mphip_warp_sample_box(_ptr(coords), _ptr(box), b, d, h, w, _stream()), "mphip_warp_sample_box")
    return box


def conv3d_roi(x: torch.Tensor, pc: "PackedConv", box: torch.Tensor, x_range: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3d evaluated only on the output tiles a sample box touches (mphip_conv3d_fwd_roi).  Every other voxel of the result
    is UNINITIALISED: the only legitimate consumer is the gather the boxes were computed for."""
    x = _req(x, "x")
    n, ci, d, h, w = x.shape
    lib = _lib.load()
    prec = _default_precision
    if prec != 0 and not lib.mphip_conv3d_supported(n, ci, pc.co, d, h, w, pc.k, prec):
        prec = 0
    wp = pc.packed(prec)
    xr = _range_for(x, x_range) if prec == 1 else None
    ws_bytes = lib.mphip_conv3d_roi_workspace_bytes(n, ci, pc.co, d, h, w, pc.k, prec)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x.device) if ws_bytes else None
    y = torch.empty((n, pc.co, d, h, w), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphip_conv3d_fwd_roi(_ptr(x), _ptr(xr), _ptr(wp), _ptr(pc.bias), _ptr(y), _ptr(box), 0, n, ci, pc.co, d, h, w, pc.k, prec,
                                        _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_fwd_roi")
    return y


# ------------------------------------------------------------------ K4 / K5
# Conv precision: 0 = exact fp32 MFMA; 1 = "f16x3" (split-f16, 3 MFMAs per product, fp32-class
# accuracy, ~3x faster).  "auto" uses f16x3 wherever the kernel supports the shape (all 3x3x3 convs
# of G3d) and exact fp32 elsewhere.  Override with MPHIP_CONV_PRECISION=fp32|f16x3|auto.
_PRECISION_NAMES = {"fp32": 0, "exact": 0, "0": 0, "f16x3": 1, "1": 1, "auto": 1}
_default_precision = _PRECISION_NAMES.get(_os.environ.get("MPHIP_CONV_PRECISION", "auto").lower(), 1)


def set_conv_precision(mode) -> None:
    """mode: 'fp32' | 'f16x3' | 'auto' (or 0/1)."""
    global _default_precision
    _default_precision = _PRECISION_NAMES[str(mode).lower()]


def get_conv_precision() -> int:
    return _default_precision


def autocast_half() -> bool:
    """True inside a torch.autocast(device_type='cuda', dtype=torch.float16) region — the reference's generator step (train.py:145,188)."""
    try:
        return bool(torch.is_autocast_enabled("cuda")) and torch.get_autocast_dtype("cuda") == torch.float16
    except TypeError:   # older torch: no device argument
        return bool(torch.is_autocast_enabled()) and torch.get_autocast_gpu_dtype() == torch.float16


def finish_graph_capture(graph: "torch.cuda.CUDAGraph") -> int:
    """For a torch.cuda.CUDAGraph(keep_graph=True) right after its capture: rewrites every MEMSET node as a kernel node
    (mphip_graph_memsets_to_kernels — on ROCm 7.x memset nodes are not reliably ordered with their neighbours, see
    training.GraphedTrainStep) and instantiates the graph.  Returns the number of nodes replaced.  MPHIP_GRAPH_MEMSET_FIX=0 (dev A/B)
    leaves the memsets alone."""
    n = ctypes.c_int(0)
    if _os.environ.get("MPHIP_GRAPH_MEMSET_FIX", "1") != "0":
        lib, left = _lib.load(), ctypes.c_int(0)
        _lib.check(lib.mphip_graph_memsets_to_kernels(graph.raw_cuda_graph(), ctypes.byref(n)), "mphip_graph_memsets_to_kernels")
        _lib.check(lib.mphip_graph_memset_nodes_left(graph.raw_cuda_graph(), ctypes.byref(left)), "mphip_graph_memset_nodes_left")
        if left.value:
            import warnings

            warnings.warn(f"finish_graph_capture: {left.value} MEMSET node(s) could not be rewritten as kernel nodes (2-D memsets, odd element "
                          "sizes or child graphs); on ROCm 7.x they are not reliably ordered with their neighbours — results they guard "
                          "(multi-block reductions) may be stale after a replay", RuntimeWarning, stacklevel=2)
    graph.instantiate()
    return n.value


def half_products_active() -> bool:
    """The calling thread's conv arithmetic policy as the library sees it (set by `half_products` / model._autocast_policy).  This — not
    torch's autocast state — is what an autograd Function must record in its forward: custom_fwd(cast_inputs=...) runs the forward with
    autocast disabled, so torch.is_autocast_enabled() is False there even inside the region."""
    lib = _lib.load()
    prev = lib.mphip_conv3d_set_half_products(0)
    lib.mphip_conv3d_set_half_products(prev)
    return bool(prev)


class half_products:
    """Context manager: the calling thread's F(2,3) conv launches and 3x3x3 f16x3 bwd-weight launches use ONE f16 product per multiply (include/mphip.h:
    mphip_conv3d_set_half_products) — the arithmetic torch.autocast(float16) gives the reference's conv3d calls.  The flag is
    thread-local in the library; nesting restores the previous value.  `enable=None`: follow torch's autocast state."""

    def __init__(self, enable: Optional[bool] = None):
        self.enable = autocast_half() if enable is None else bool(enable)

    def __enter__(self):
        self.prev = _lib.load().mphip_conv3d_set_half_products(int(self.enable))
        return self

    def __exit__(self, *exc):
        _lib.load().mphip_conv3d_set_half_products(self.prev)
        return False


def f16x3_saturation_count(reset: bool = False) -> int:
    """Operand elements of the f16x3 conv kernels whose scaled value was outside the f16 range since the last reset: Inf /
    NaN inputs (or finite values beyond a stale, hand-supplied range descriptor).  They are not clamped — the result
    carries Inf/NaN like the reference's fp32 conv — this only counts them.  0 in normal operation.
    Synchronises the device (a diagnostic, not for the hot loop)."""
    c = ctypes.c_ulonglong(0)
    _lib.check(_lib.load().mphip_f16x3_saturation_count(ctypes.byref(c), int(reset)), "mphip_f16x3_saturation_count")
    return int(c.value)


class PackedConv:
    """One Conv3d / 1x1 Conv2d: OIDHW fp32 weight + bias, packed lazily per precision into the
    kernel layouts described in include/mphip.h."""

    __slots__ = ("weight", "bias", "co", "ci", "k", "_packed", "transposed", "header_from", "_table_token")

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], transposed: bool = False,
                 header_from: Optional["PackedConv"] = None):
        """transposed: this object is the bwd-data conv of the conv whose weight is `weight` (Co/Ci swapped, taps
        flipped); the packing kernels read the original layout directly."""
        weight = _req(weight.detach(), "conv weight")
        self.transposed = bool(transposed)
        self.header_from = header_from  # (transposed packs) the forward PackedConv of the same weight: shares its f16x3 scale
        co, ci = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        k = weight.shape[2]
        if weight.dim() == 4:  # Conv2d 1x1 (model.py:425)
            if tuple(weight.shape[2:]) != (1, 1):
                raise RuntimeError("PackedConv: only 1x1 Conv2d is on the hot path")
        elif weight.dim() != 5 or tuple(weight.shape[2:]) != (k, k, k) or k not in (1, 3):
            raise RuntimeError(f"PackedConv: unsupported weight shape {tuple(weight.shape)}")
        self.weight = weight
        self.bias = None if bias is None else _req(bias.detach(), "conv bias")  # a view: the owner re-creates the pack when it changes
        self.co, self.ci, self.k = co, ci, k
        self._packed = {}
        self._table_token = -1   # ops.PackTable: the repack_always region whose table run wrote these packs

    def packed(self, precision: int) -> torch.Tensor:
        wp = self._packed.get(precision)
        if wp is None:
            lib = _lib.load()
            nbytes = lib.mphip_packed_weight_bytes(self.co, self.ci, self.k, precision)
            if nbytes == 0:
                raise RuntimeError(f"PackedConv: precision {precision} not available for Co={self.co} Ci={self.ci} k={self.k}")
            wp = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.weight.device)
            src = self.header_from._packed.get(1) if (self.transposed and precision == 1 and self.header_from is not None) else None
            if src is not None:
                _lib.check(lib.mphip_pack_conv_weight_bwd_data_like(_ptr(self.weight), _ptr(wp), self.co, self.ci, self.k, precision,
                                                                    _ptr(src), _stream()), "mphip_pack_conv_weight_bwd_data_like")
            else:
                fn = lib.mphip_pack_conv_weight_bwd_data if self.transposed else lib.mphip_pack_conv_weight
                _lib.check(fn(_ptr(self.weight), _ptr(wp), self.co, self.ci, self.k, precision, _stream()), "mphip_pack_conv_weight")
            self._packed[precision] = wp
        return wp

    @property
    def wp(self) -> torch.Tensor:  # exact-fp32 packing (kept for tools/tests)
        return self.packed(0)


_conv_hook = None
_repack_always = False
_repack_token = 0      # bumped by every repack_always region: a PackTable run inside one marks its packs with it (pack_is_current)


class repack_always:
    """Context: packed-weight caches are bypassed (training.GraphedTrainStep captures the re-packing of the updated
    weights into its graph; a cache hit at capture time would freeze stale packs into every replay)."""

    def __enter__(self):
        global _repack_always, _repack_token
        with _state_lock:
            self._old, _repack_always = _repack_always, True
            _repack_token += 1

    def __exit__(self, *exc):
        global _repack_always
        _repack_always = self._old


def repacking() -> bool:
    return _repack_always


def pack_is_current(hit, key) -> bool:
    """The cache test of every per-module pack cache (model._PackCache, autograd._bwd_pack): the cached (key, PackedConv) pair serves
    `key` — and, inside a repack_always region (a hipGraph capture), only if a PackTable run of THIS region wrote it (the capture must
    contain the re-packing of every weight it uses; a pack made before it would freeze stale values into every replay)."""
    if hit is None or hit[0] != key:
        return False
    return (not _repack_always) or getattr(hit[1], "_table_token", -1) == _repack_token


class PackTable:
    """Every weight pack a module's training step needs, re-made in at most five launches (mphip_pack_table_*; bit-identical to the
    lazy per-conv packs).  Build it AFTER one real step has run — `PackTable.from_module(model)` collects the forward packs
    (model._PackCache) and bwd-data packs (autograd._bwd_pack) that step created, with the precisions it used — then call `run()` at the
    top of every step, after the optimizer update and before the forward: it re-packs all of them from the current weights and
    re-validates the module caches, so that the step's convs find their packs current.  Convs the warm-up step did not touch fall back
    to the lazy path as before; a precision a covered conv acquires later (a shape outside the fast kernel's tiling -> precision 0) is
    dropped by every run and re-packed lazily from the current weights.  training.GraphedTrainStep does all of this by itself."""

    def __init__(self, entries):
        """entries: (conv module, cache attribute, PackedConv) triples; every PackedConv contributes the precisions it holds."""
        lib = _lib.load()
        self._entries = list(entries)
        # the precisions each PackedConv held when the table was built = what a run re-packs; anything added later is NOT covered
        self._covered = [frozenset(pc._packed) for _, _, pc in self._entries]
        jobs = []
        fwd_of = {}   # weight storage -> the forward f16x3 pack buffer of the same weight (its header is shared)
        for conv, attr, pc in self._entries:
            if not pc.transposed and 1 in pc._packed:
                fwd_of[pc.weight.data_ptr()] = pc._packed[1]
        for conv, attr, pc in self._entries:
            for prec, wp in sorted(pc._packed.items()):
                like = fwd_of.get(pc.weight.data_ptr()) if (pc.transposed and prec == 1) else None
                jobs.append(_lib.PackJob(pc.weight.data_ptr(), wp.data_ptr(), None if like is None else like.data_ptr(), pc.co, pc.ci, pc.k,
                                         prec, int(pc.transposed)))
        if not jobs:
            raise RuntimeError("PackTable: nothing to pack (run one step of the module first)")
        self.n_jobs = len(jobs)
        arr = (_lib.PackJob * len(jobs))(*jobs)
        handle = ctypes.c_void_p()
        _lib.check(lib.mphip_pack_table_create(ctypes.cast(arr, ctypes.c_void_p), len(jobs), ctypes.byref(handle)), "mphip_pack_table_create")
        self._handle = handle

    @classmethod
    def from_module(cls, module: "torch.nn.Module") -> "PackTable":
        entries = []
        for m in module.modules():
            for attr in ("_mphip_pack", "_mphip_bwd_pack"):
                hit = m.__dict__.get(attr)
                if hit is not None and isinstance(hit[1], PackedConv) and hit[1]._packed:
                    entries.append((m, attr, hit[1]))
        return cls(entries)

    def run(self) -> None:
        for conv, attr, pc in self._entries:   # the table holds raw pointers: a parameter that moved (module.to(), a replaced Parameter) makes it stale
            if conv.weight.data_ptr() != pc.weight.data_ptr():
                raise RuntimeError("PackTable.run: a conv weight moved since the table was built (module.to() / a replaced Parameter); "
                                   "build a new table with PackTable.from_module")
        if not self._handle:
            raise RuntimeError("PackTable.run: the table was closed")
        _lib.check(_lib.load().mphip_pack_table_run(self._handle, _stream()), "mphip_pack_table_run")
        for (conv, attr, pc), covered in zip(self._entries, self._covered):
            # a precision packed lazily AFTER the table was built (a ragged last batch falling back to precision 0,
            # set_conv_precision) is not among the table's jobs: drop it so that packed() re-makes it from the current weights
            for prec in [q for q in pc._packed if q not in covered]:
                del pc._packed[prec]
            pc._table_token = _repack_token
            pc.weight = conv.weight.detach()
            pc.bias = None if (pc.transposed or conv.bias is None) else conv.bias.detach()
            conv.__dict__[attr] = (_pack_cache_key(conv, attr), pc)

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().mphip_pack_table_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _pack_cache_key(conv, attr):
    """The cache keys of model._PackCache.get / autograd._bwd_pack (kept here so that PackTable.run can re-validate both)."""
    w, b = conv.weight, conv.bias
    if attr == "_mphip_pack":
        return (w.data_ptr(), w._version, tuple(w.shape), None if b is None else (b.data_ptr(), b._version), str(w.device), weight_epoch())
    return (w.data_ptr(), w._version, tuple(w.shape), str(w.device), weight_epoch())


_weight_epoch = 0


def invalidate_packs() -> None:
    """Bumps the epoch that is part of every packed-weight cache key.  MANDATORY after any write that changes parameter
    memory without bumping the tensors' autograd version counters: `param.data.copy_/mul_/clamp_` (EMA, weight clipping,
    manual loads through `.data`), external graph replays — the caches key on (data_ptr, _version, epoch) and would
    otherwise keep serving the old packed weights.  Ordinary in-place updates (optimizers, load_state_dict, `with
    torch.no_grad(): p.copy_(...)`) bump `_version` and need nothing.  training.GraphedTrainStep calls this after
    every replay."""
    global _weight_epoch
    with _state_lock:
        _weight_epoch += 1


def weight_epoch() -> int:
    return _weight_epoch


def set_conv_hook(hook) -> None:
    """Measurement hook (bench.py): hook(x, packed_conv, launch) -> y wraps each conv launch, e.g. to
    bracket the dominant kernel with HIP events on the launch stream.  None disables it."""
    global _conv_hook
    _conv_hook = hook


def conv3d(x: torch.Tensor, pc: PackedConv, precision: Optional[int] = None, gn_groups: Optional[int] = None,
           gn_eps: float = 1e-5, x_range: Optional[torch.Tensor] = None):
    """y = conv(x).  With gn_groups, also returns the (mean, rstd) statistics of y for the GroupNorm that
    follows (one fused pass for split-K convs): -> (y, stats)."""
    x = _req(x, "x")
    if x.dim() != 5 or x.shape[1] != pc.ci:
        raise RuntimeError(f"conv3d: input {tuple(x.shape)} does not match Ci={pc.ci}")
    n, ci, d, h, w = x.shape
    lib = _lib.load()
    prec = _default_precision if precision is None else precision
    if prec != 0 and not lib.mphip_conv3d_supported(n, ci, pc.co, d, h, w, pc.k, prec):
        prec = 0  # shape outside the fast kernel's tiling: the exact fp32 kernel handles every shape
    wp = pc.packed(prec)
    xr = _range_for(x, x_range) if prec == 1 else None   # f16x3: the input's own operand scale
    if gn_groups:
        ws_bytes = lib.mphip_conv3d_gn_workspace_bytes(n, ci, pc.co, d, h, w, pc.k, prec, gn_groups)
        stats = torch.empty((n * gn_groups, 2), dtype=torch.float32, device=x.device)
    else:
        ws_bytes = lib.mphip_conv3d_workspace_bytes(n, ci, pc.co, d, h, w, pc.k, prec)
        stats = None
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x.device) if ws_bytes else None
    y = torch.empty((n, pc.co, d, h, w), dtype=torch.float32, device=x.device)

    def launch():
        if gn_groups:
            _lib.check(lib.mphip_conv3d_gn_fwd(_ptr(x), _ptr(xr), _ptr(wp), _ptr(pc.bias), _ptr(y), _ptr(stats), n, ci, pc.co, d, h, w,
                                               pc.k, prec, gn_groups, gn_eps, _ptr(ws), ws_bytes, _stream()),
                       "mphip_conv3d_gn_fwd")
        else:
            _lib.check(lib.mphip_conv3d_fwd(_ptr(x), _ptr(xr), _ptr(wp), _ptr(pc.bias), _ptr(y), n, ci, pc.co, d, h, w, pc.k, prec,
                                            _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_fwd")
        return y

    y = _conv_hook(x, pc, launch) if _conv_hook is not None else launch()
    return (y, stats) if gn_groups else y


_GNIN_ENABLED = _os.environ.get("MPHIP_GN_IN_CONV", "1") != "0"  # dev switch for same-box A/B runs


def gn_in_conv_ok(x_shape, pc: "PackedConv") -> bool:
    """True when `conv(relu(groupnorm(x)))` can run as ONE conv launch (f16x3 kernel with the norm folded into its
    input staging)."""
    if not _GNIN_ENABLED or _default_precision != 1 or pc.k != 3 or pc.ci > 768:
        return False
    n, ci, d, h, w = x_shape
    return bool(_lib.load().mphip_conv3d_supported(n, ci, pc.co, d, h, w, pc.k, 1))


def conv3d_gn_in(x: torch.Tensor, stats: torch.Tensor, gamma, beta, groups: int, pc: "PackedConv", w2=None, b2=None,
                 relu: bool = True, out_gn_groups: Optional[int] = None, out_gn_eps: float = 1e-5):
    """conv(relu(GN(x))) with the normalisation applied inside the conv's input staging (mphip_conv3d_gnin_fwd).
    With out_gn_groups, also returns the (mean, rstd) statistics of the result for the next GroupNorm: -> (y, stats)."""
    x = _req(x, "x")
    n, ci, d, h, w = x.shape
    lib = _lib.load()
    gamma, beta = _req(gamma.detach(), "gamma"), _req(beta.detach(), "beta")
    if w2 is not None:
        w2, b2 = _req(w2.detach(), "w2").reshape(-1), _req(b2.detach(), "b2").reshape(-1)
    table = torch.empty((n, ci, 2), dtype=torch.float32, device=x.device)
    xr = new_range(x.device)  # range of the NORMALISED tensor the conv sees (a data-independent bound, see include/mphip.h)
    _lib.check(lib.mphip_groupnorm_affine_table(_ptr(stats), _ptr(gamma), _ptr(beta), _ptr(w2), _ptr(b2), _ptr(table), _ptr(xr), n,
                                                ci, d * h * w, groups, _stream()), "mphip_groupnorm_affine_table")
    wp = pc.packed(1)
    if out_gn_groups:
        ws_bytes = lib.mphip_conv3d_gn_workspace_bytes(n, ci, pc.co, d, h, w, pc.k, 1, out_gn_groups)
        out_stats = torch.empty((n * out_gn_groups, 2), dtype=torch.float32, device=x.device)
    else:
        ws_bytes = lib.mphip_conv3d_workspace_bytes(n, ci, pc.co, d, h, w, pc.k, 1)
        out_stats = None
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x.device) if ws_bytes else None
    y = torch.empty((n, pc.co, d, h, w), dtype=torch.float32, device=x.device)

    def launch():
        if out_gn_groups:
            _lib.check(lib.mphip_conv3d_gnin_gn_fwd(_ptr(x), _ptr(table), _ptr(xr), int(relu), _ptr(wp), _ptr(pc.bias), _ptr(y), _ptr(out_stats),
                                                    n, ci, pc.co, d, h, w, pc.k, 1, out_gn_groups, out_gn_eps, _ptr(ws), ws_bytes,
                                                    _stream()), "mphip_conv3d_gnin_gn_fwd")
        else:
            _lib.check(lib.mphip_conv3d_gnin_fwd(_ptr(x), _ptr(table), _ptr(xr), int(relu), _ptr(wp), _ptr(pc.bias), _ptr(y), n, ci, pc.co, d,
                                                 h, w, pc.k, 1, _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_gnin_fwd")
        return y

    y = _conv_hook(x, pc, launch) if _conv_hook is not None else launch()
    return (y, out_stats) if out_gn_groups else y


class ConvOut:
    """A conv result that may still be in split-K form: `data` is [splits, N, Co, D, H, W] partial slabs (bias
    not added, passed on in `bias`) when splits > 1, or the finished [N, Co, D, H, W] tensor when splits == 1."""

    __slots__ = ("data", "splits", "bias", "shape", "stats", "stats_groups")

    def __init__(self, data, splits, bias, shape, stats=None, stats_groups=0):
        self.data, self.splits, self.bias, self.shape = data, splits, bias, shape
        self.stats, self.stats_groups = stats, stats_groups  # GroupNorm (mean, rstd) of the result, when the conv produced them


def conv3d_split(x: torch.Tensor, pc: PackedConv, precision: Optional[int] = None, gn_groups: Optional[int] = None,
                 gn_eps: float = 1e-5) -> ConvOut:
    """Conv whose split-K reduction (small volumes) is left to the GroupNorm kernels that consume it.  gn_groups: the
    result feeds a GroupNorm — a direct (unsplit) launch then carries the statistics along (ConvOut.stats: computed in
    the f16x3 kernel's epilogue instead of a separate pass over the tensor)."""
    x = _req(x, "x")
    if x.dim() != 5 or x.shape[1] != pc.ci:
        raise RuntimeError(f"conv3d: input {tuple(x.shape)} does not match Ci={pc.ci}")
    n, ci, d, h, w = x.shape
    lib = _lib.load()
    prec = _default_precision if precision is None else precision
    if prec != 0 and not lib.mphip_conv3d_supported(n, ci, pc.co, d, h, w, pc.k, prec):
        prec = 0
    splits = lib.mphip_conv3d_splits(n, ci, pc.co, d, h, w, pc.k, prec)
    shape = (n, pc.co, d, h, w)
    elems = n * pc.co * d * h * w
    if splits > 1 and (elems > _SPLIT_CHAIN_MAX_ELEMS or (splits >= _SPLIT_CHAIN_MAX_SPLITS and splits * elems > _SPLIT_CHAIN_MAX_SLAB_ELEMS)):
        # mid-sized tensors (G3d's inner levels): the dedicated reduce + vectorised GN kernels are faster than the
        # one-thread-per-element split-aware kernels (measured: -13 % end to end when those were used everywhere)
        if gn_groups:
            y, st = conv3d(x, pc, precision=prec, gn_groups=gn_groups, gn_eps=gn_eps)
            return ConvOut(y, 1, None, shape, st, gn_groups)
        return ConvOut(conv3d(x, pc, precision=prec), 1, None, shape)
    if gn_groups and splits == 1 and prec == 1:
        y, st = conv3d(x, pc, precision=prec, gn_groups=gn_groups, gn_eps=gn_eps)
        return ConvOut(y, 1, None, shape, st, gn_groups)
    out = torch.empty((splits,) + shape if splits > 1 else shape, dtype=torch.float32, device=x.device)
    wp = pc.packed(prec)
    xr = _range_for(x) if prec == 1 else None

    def launch():
        _lib.check(lib.mphip_conv3d_fwd_split(_ptr(x), _ptr(xr), _ptr(wp), _ptr(pc.bias), _ptr(out), n, ci, pc.co, d, h, w, pc.k,
                                              prec, None, 0, _stream()), "mphip_conv3d_fwd_split")
        return out

    if _conv_hook is not None:
        _conv_hook(x, pc, launch)
    else:
        launch()
    return ConvOut(out, splits, pc.bias if splits > 1 else None, shape)


_SPLIT_CHAIN_MAX_ELEMS = 1 << 18  # conv outputs up to 1 MB keep their split-K slabs for the GN kernels (FlowField)
# ... unless there are many of them (a single frame's 2x8x8 level: 24-48 slabs, 9-19 MB): the split-aware statistics kernel is one
# workgroup per (sample, group) — 32 workgroups at B=1, each reading every slab of its channels (16-23 us) — while the ordered reduce
# uses the whole chip (r04: B=1 step -60 us)
_SPLIT_CHAIN_MAX_SPLITS = 16
_SPLIT_CHAIN_MAX_SLAB_ELEMS = 1 << 20
_STATS_SPLIT_MAX_SPAN = 65536  # floats per (sample, group) the single-launch split-aware statistics kernel accepts


def _finish(co: ConvOut) -> torch.Tensor:
    """Plain tensor from a ConvOut (sums the slabs with torch when a consumer needs the finished tensor)."""
    if co.splits == 1:
        return co.data
    y = co.data.sum(dim=0)
    return y if co.bias is None else y + co.bias.view(1, -1, 1, 1, 1)


# ------------------------------------------------------------------ K6
def groupnorm_stats(x, groups: int, eps: float = 1e-5) -> torch.Tensor:
    lib = _lib.load()
    if isinstance(x, ConvOut) and x.stats is not None and x.stats_groups == groups:
        return x.stats  # produced by the conv itself
    if isinstance(x, ConvOut):
        n, c, d, h, w = x.shape
        s = d * h * w
        if x.splits > 1 and (c // groups) * s <= _STATS_SPLIT_MAX_SPAN:
            stats = torch.empty((n * groups, 2), dtype=torch.float32, device=x.data.device)
            _lib.check(lib.mphip_groupnorm_stats_split(_ptr(x.data), x.splits, _ptr(x.bias), _ptr(stats), n, c, s, groups, eps,
                                                       _stream()), "mphip_groupnorm_stats_split")
            return stats
        x = _finish(x)
    x = _req(x, "x")
    n, c = x.shape[0], x.shape[1]
    s = x.numel() // (n * c)
    stats = torch.empty((n * groups, 2), dtype=torch.float32, device=x.device)
    ws_bytes = lib.mphip_groupnorm_workspace_bytes(n, c, s, groups)
    if ws_bytes == 0:
        raise RuntimeError(f"groupnorm_stats: bad dims C={c} G={groups}")
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=x.device)
    _lib.check(lib.mphip_groupnorm_stats(_ptr(x), _ptr(stats), n, c, s, groups, eps, _ptr(ws), ws_bytes, _stream()),
               "mphip_groupnorm_stats")
    return stats


def groupnorm_apply(x, stats, gamma, beta, groups: int, w2=None, b2=None, residual=None, relu=False, tanh=False,
                    pool2=False, up=(1, 1, 1)) -> torch.Tensor:
    """GroupNorm apply (+second affine, +residual, +ReLU, +tanh, then 2x2x2 average pool or nearest upsample).
    `x` / `residual` may be ConvOut objects still in split-K form."""
    lib = _lib.load()
    xs = x.splits if isinstance(x, ConvOut) else 1
    rs = residual.splits if isinstance(residual, ConvOut) else 1
    xt = x.data if isinstance(x, ConvOut) else x
    rt = residual.data if isinstance(residual, ConvOut) else residual
    shape = x.shape if isinstance(x, ConvOut) else tuple(x.shape)
    if len(shape) != 5:
        raise RuntimeError("groupnorm_apply: expected a 5-D NCDHW tensor")
    n, c, d, h, w = shape
    xt = _req(xt, "x")
    gamma, beta = _req(gamma.detach(), "gamma"), _req(beta.detach(), "beta")
    if w2 is not None:
        w2, b2 = _req(w2.detach(), "w2").reshape(-1), _req(b2.detach(), "b2").reshape(-1)
    if rt is not None:
        rt = _req(rt, "residual")
        rshape = residual.shape if isinstance(residual, ConvOut) else tuple(residual.shape)
        if tuple(rshape) != tuple(shape):
            raise RuntimeError("groupnorm_apply: residual shape mismatch")
    up = tuple(int(u) for u in up)
    general = xs > 1 or rs > 1 or up != (1, 1, 1)
    if pool2:
        oshape = (n, c, d // 2, h // 2, w // 2)
    else:
        oshape = (n, c, d * up[0], h * up[1], w * up[2])
    y = torch.empty(oshape, dtype=torch.float32, device=xt.device)
    rng = new_range(xt.device) if _RANGES_ENABLED else None  # max|y| rides along: the next f16x3 conv's operand scale
    if general:
        xb = x.bias if isinstance(x, ConvOut) else None
        rb = residual.bias if isinstance(residual, ConvOut) else None
        _lib.check(lib.mphip_groupnorm_apply_split(_ptr(xt), xs, _ptr(xb), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(w2), _ptr(b2),
                                                   _ptr(rt), rs, _ptr(rb), _ptr(y), _ptr(rng), n, c, d, h, w, groups, int(relu),
                                                   int(tanh), int(pool2), up[0], up[1], up[2], _stream()),
                   "mphip_groupnorm_apply_split")
    else:
        _lib.check(lib.mphip_groupnorm_apply(_ptr(xt), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(w2), _ptr(b2), _ptr(rt),
                                             _ptr(y), _ptr(rng), n, c, d, h, w, groups, int(relu), int(tanh), int(pool2),
                                             _stream()), "mphip_groupnorm_apply")
    return tag_range(y, rng)


_GN_FUSED_MAX_SPAN = 12288  # floats per (sample, group): LDS cache of the one-launch GroupNorm


_GN_SMALL_ENABLED = _os.environ.get("MPHIP_GN_SMALL", "1") != "0"  # dev switch for same-box A/B runs


def groupnorm_fused_ok(x, groups: int) -> bool:
    if not _GN_SMALL_ENABLED:
        return False
    shape = x.shape if isinstance(x, ConvOut) else tuple(x.shape)
    n, c, d, h, w = shape
    return (c // groups) * d * h * w <= _GN_FUSED_MAX_SPAN


def groupnorm_small(x, gamma, beta, groups: int, eps: float = 1e-5, w2=None, b2=None, residual=None, relu=False,
                    tanh=False, up=(1, 1, 1)) -> torch.Tensor:
    """GroupNorm statistics + apply (+second affine, residual, ReLU, tanh, nearest upsample) in one launch for tiny
    tensors; `x` / `residual` may be split-K ConvOut objects."""
    lib = _lib.load()
    xs = x.splits if isinstance(x, ConvOut) else 1
    rs = residual.splits if isinstance(residual, ConvOut) else 1
    xt = _req(x.data if isinstance(x, ConvOut) else x, "x")
    rt = residual.data if isinstance(residual, ConvOut) else residual
    shape = x.shape if isinstance(x, ConvOut) else tuple(x.shape)
    n, c, d, h, w = shape
    gamma, beta = _req(gamma.detach(), "gamma"), _req(beta.detach(), "beta")
    if w2 is not None:
        w2, b2 = _req(w2.detach(), "w2").reshape(-1), _req(b2.detach(), "b2").reshape(-1)
    if rt is not None:
        rt = _req(rt, "residual")
    xb = x.bias if isinstance(x, ConvOut) else None
    rb = residual.bias if isinstance(residual, ConvOut) else None
    up = tuple(int(u) for u in up)
    y = torch.empty((n, c, d * up[0], h * up[1], w * up[2]), dtype=torch.float32, device=xt.device)
    _lib.check(lib.mphip_groupnorm_small_fused(_ptr(xt), xs, _ptr(xb), _ptr(gamma), _ptr(beta), _ptr(w2), _ptr(b2), _ptr(rt), rs,
                                               _ptr(rb), _ptr(y), None, n, c, d, h, w, groups, eps, int(relu), int(tanh), up[0],
                                               up[1], up[2], _stream()), "mphip_groupnorm_small_fused")
    return y


_FF_FUSED = _os.environ.get("MPHIP_FF_FUSED", "1") != "0"  # dev switch for same-box A/B runs (the plan reads the same variable)


def flowfield_conv_gn_ok(x_shape, conv, res_conv=None, groups: int = 32) -> bool:
    """True when `relu(AGN(conv(x)) [+ res_conv(xr)])` is one of FlowField's block halves (mphip_flowfield_conv_gn)."""
    if not _FF_FUSED or len(x_shape) != 5:
        return False
    n, ci, d, h, w = x_shape
    cr = res_conv.weight.shape[1] if res_conv is not None else 0
    return bool(_lib.load().mphip_flowfield_conv_gn_supported(ci, conv.weight.shape[0], d, h, w, cr, groups))


def flowfield_conv_gn(x, conv, norm, res_x=None, res_conv=None, relu=True, up=(1, 1, 1)) -> torch.Tensor:
    """One launch: upsample_nearest(relu(AGN(conv3x3x3(x)) [+ conv1x1x1(res_x)])) for FlowField's four levels (model.py:369-408 at
    439-471), from the modules' own weight tensors.  `norm`: AdaptiveGroupNorm (group_norm.{weight,bias}, weight, bias)."""
    x = _req(x, "x")
    n, ci, d, h, w = x.shape
    co = conv.weight.shape[0]
    wt, bs = _req(conv.weight.detach(), "conv.weight"), (_req(conv.bias.detach(), "conv.bias") if conv.bias is not None else None)
    g, b = _req(norm.group_norm.weight.detach(), "gamma"), _req(norm.group_norm.bias.detach(), "beta")
    w2, b2 = _req(norm.weight.detach(), "w2").reshape(-1), _req(norm.bias.detach(), "b2").reshape(-1)
    rx = rw = rb = None
    cr = 0
    if res_conv is not None:
        rx = _req(res_x, "res_x")
        rw = _req(res_conv.weight.detach(), "res_conv.weight")
        rb = _req(res_conv.bias.detach(), "res_conv.bias") if res_conv.bias is not None else None
        cr = rw.shape[1]
    up = tuple(int(u) for u in up)
    y = torch.empty((n, co, d * up[0], h * up[1], w * up[2]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mphip_flowfield_conv_gn(_ptr(x), _ptr(wt), _ptr(bs), _ptr(g), _ptr(b), _ptr(w2), _ptr(b2), _ptr(rx), _ptr(rw),
                                                   _ptr(rb), _ptr(y), n, ci, co, d, h, w, cr, up[0], up[1], up[2], norm.num_groups,
                                                   norm.group_norm.eps, int(relu), _stream()), "mphip_flowfield_conv_gn")
    return y


def flowfield_out_ok(x_shape, conv) -> bool:
    return _FF_FUSED and tuple(x_shape[1:]) == (32, 16, 16, 16) and tuple(conv.weight.shape) == (3, 32, 3, 3, 3)


def flowfield_out(x, conv, gn) -> torch.Tensor:
    """tanh(relu(GroupNorm(1, 3)(conv3x3x3(x)))) for FlowField's output head (model.py:458-465): two launches."""
    x = _req(x, "x")
    n = x.shape[0]
    lib = _lib.load()
    wt, bs = _req(conv.weight.detach(), "conv.weight"), (_req(conv.bias.detach(), "conv.bias") if conv.bias is not None else None)
    g, b = _req(gn.weight.detach(), "gamma"), _req(gn.bias.detach(), "beta")
    ws_bytes = lib.mphip_flowfield_out_workspace_bytes(n)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x.device)
    em = torch.empty((n, 3, 16, 16, 16), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphip_flowfield_out(_ptr(x), _ptr(wt), _ptr(bs), _ptr(g), _ptr(b), _ptr(em), n, gn.eps, _ptr(ws), ws_bytes, _stream()),
               "mphip_flowfield_out")
    return em


# ------------------------------------------------------------------ K7
def avgpool2(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, "x")
    n, c, d, h, w = x.shape
    y = torch.empty((n, c, d // 2, h // 2, w // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mphip_avgpool2(_ptr(x), _ptr(y), n * c, d, h, w, _stream()), "mphip_avgpool2")
    return tag_range(y, tensor_range(x))  # an average: max|y| <= max|x|


def upsample_trilinear2(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, "x")
    n, c, d, h, w = x.shape
    y = torch.empty((n, c, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mphip_upsample_trilinear2(_ptr(x), _ptr(y), n * c, d, h, w, _stream()),
               "mphip_upsample_trilinear2")
    return tag_range(y, tensor_range(x))  # convex combinations of x: max|y| <= max|x|


def _int_scale(scale) -> Tuple[int, int, int]:
    out = tuple(int(round(float(v))) for v in scale)
    if len(out) != 3 or any(o < 1 or abs(float(v) - o) > 1e-9 for o, v in zip(out, scale)):
        raise ValueError(f"upsample_trilinear: integer scale factors (sD,sH,sW) >= 1 expected, got {tuple(scale)}")
    return out


def upsample_trilinear(x: torch.Tensor, scale) -> torch.Tensor:
    """F.interpolate(x, scale_factor=scale, mode='trilinear', align_corners=False) (model.py:404-405, 525-526)."""
    x = _req(x, "x")
    n, c, d, h, w = x.shape
    sd, sh, sw = _int_scale(scale)
    y = torch.empty((n, c, d * sd, h * sh, w * sw), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mphip_upsample_trilinear(_ptr(x), _ptr(y), n * c, d, h, w, sd, sh, sw, _stream()), "mphip_upsample_trilinear")
    return tag_range(y, tensor_range(x))  # convex combinations of x


def upsample_trilinear_bwd(dout: torch.Tensor, scale) -> torch.Tensor:
    dout = _req(dout, "dout")
    n, c, d, h, w = dout.shape
    sd, sh, sw = _int_scale(scale)
    if d % sd or h % sh or w % sw:
        raise RuntimeError(f"upsample_trilinear_bwd: gradient dims {(d, h, w)} are not multiples of the scale factors {(sd, sh, sw)}")
    dx = torch.empty((n, c, d // sd, h // sh, w // sw), dtype=torch.float32, device=dout.device)
    if dx.numel() % 4:
        # (the kernel zero-fills dx with 16-byte stores before its fp32 atomic scatter: not bitwise reproducible run to run,
        #  unlike the x2 align_corners=True adjoint G3d uses — no module of Gbase sets `upsample=True`, model.py:404,525)
        raise RuntimeError(f"upsample_trilinear_bwd: input-gradient size {dx.numel()} must be a multiple of 4 elements")
    _lib.check(_lib.load().mphip_upsample_trilinear_bwd(_ptr(dout), _ptr(dx), n * c, d // sd, h // sh, w // sw, sd, sh, sw, _stream()),
               "mphip_upsample_trilinear_bwd")
    return dx


def upsample_nearest(x: torch.Tensor, scale: Tuple[int, int, int]) -> torch.Tensor:
    x = _req(x, "x")
    n, c, d, h, w = x.shape
    sd, sh, sw = scale
    y = torch.empty((n, c, d * sd, h * sh, w * sw), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mphip_upsample_nearest(_ptr(x), _ptr(y), n * c, d, h, w, sd, sh, sw, _stream()),
               "mphip_upsample_nearest")
    return tag_range(y, tensor_range(x))


# ------------------------------------------------------------------ K8
def add_matmul(a: torch.Tensor, a2: Optional[torch.Tensor], m: torch.Tensor, bias: Optional[torch.Tensor] = None,
               trans: bool = False) -> torch.Tensor:
    a = _req(a, "a")
    a2 = None if a2 is None else _req(a2, "a2")
    m = _req(m.detach(), "m")
    bias = None if bias is None else _req(bias.detach(), "bias")
    b, k = a.shape
    n = m.shape[0] if trans else m.shape[1]
    if (m.shape[1] if trans else m.shape[0]) != k:
        raise RuntimeError(f"add_matmul: inner dims differ: a={tuple(a.shape)} m={tuple(m.shape)} trans={trans}")
    out = torch.empty((b, n), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().mphip_add_matmul(_ptr(a), _ptr(a2), _ptr(m), _ptr(bias), _ptr(out), b, k, n, int(trans),
                                            _stream()), "mphip_add_matmul")
    return out


# ------------------------------------------------------------------ K9  backward (scope row f2)
def grad_prep(dy: torch.Tensor, want_bias: bool = True):
    """One pass over a conv's output gradient: (dbias [C] or None, scale) — `scale` is the device-side power-of-two
    operand scale the f16x3 backward kernels use for this tensor."""
    dy = _req(dy, "dy")
    n, c = dy.shape[0], dy.shape[1]
    s = dy.numel() // (n * c)
    scale = torch.empty(4, dtype=torch.float32, device=dy.device)
    db = torch.empty(c, dtype=torch.float32, device=dy.device) if want_bias else None
    lib = _lib.load()
    ws = torch.empty(lib.mphip_grad_prep_workspace_bytes(n, c, s) // 4, dtype=torch.float32, device=dy.device)
    _lib.check(lib.mphip_grad_prep(_ptr(dy), _ptr(db), _ptr(scale), n, c, s, _ptr(ws), ws.numel() * 4, _stream()),
               "mphip_grad_prep")
    return db, scale


def conv3d_bwd_data(dy: torch.Tensor, pc_t: "PackedConv", dy_scale: torch.Tensor, precision: Optional[int] = None,
                    roi: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx of y = conv3d(x, W): the forward kernels on the flipped/transposed weight `pc_t` (conv_bwd_data_weight).
    roi: dy is zero outside these per-frame boxes (the gradient of a gather) -> only the tiles around them are computed."""
    dy = _req(dy, "dy")
    n, ci, d, h, w = dy.shape
    if ci != pc_t.ci:
        raise RuntimeError(f"conv3d_bwd_data: dy {tuple(dy.shape)} does not match Co={pc_t.ci}")
    lib = _lib.load()
    prec = _default_precision if precision is None else precision
    if prec != 0 and not lib.mphip_conv3d_supported(n, ci, pc_t.co, d, h, w, pc_t.k, prec):
        prec = 0
    wp = pc_t.packed(prec)
    dx = torch.empty((n, pc_t.co, d, h, w), dtype=torch.float32, device=dy.device)
    if roi is not None:
        ws_bytes = lib.mphip_conv3d_roi_workspace_bytes(n, ci, pc_t.co, d, h, w, pc_t.k, prec)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=dy.device) if ws_bytes else None
        _lib.check(lib.mphip_conv3d_bwd_data_roi(_ptr(dy), _ptr(wp), _ptr(dx), _ptr(dy_scale), _ptr(roi), n, ci, pc_t.co, d, h, w, pc_t.k, prec,
                                                 _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_bwd_data_roi")
        return dx
    ws_bytes = lib.mphip_conv3d_workspace_bytes(n, ci, pc_t.co, d, h, w, pc_t.k, prec)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=dy.device) if ws_bytes else None
    _lib.check(lib.mphip_conv3d_bwd_data(_ptr(dy), _ptr(wp), _ptr(dx), _ptr(dy_scale), n, ci, pc_t.co, d, h, w, pc_t.k, prec,
                                         _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_bwd_data")
    return dx


def conv3d_bwd_weight(x: torch.Tensor, dy: torch.Tensor, k: int, dy_scale: Optional[torch.Tensor] = None,
                      precision: Optional[int] = None, x_range: Optional[torch.Tensor] = None, roi: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [Co,Ci,k,k,k] of y = conv3d(x, W, b, padding=k//2) given dy.  precision 1 (f16x3) needs dy_scale (grad_prep)."""
    x, dy = _req(x, "x"), _req(dy, "dy")
    n, ci, d, h, w = x.shape
    co = dy.shape[1]
    if tuple(dy.shape) != (n, co, d, h, w):
        raise RuntimeError(f"conv3d_bwd_weight: dy {tuple(dy.shape)} does not match x {tuple(x.shape)}")
    lib = _lib.load()
    prec = _default_precision if precision is None else precision
    if prec != 0 and (dy_scale is None or not lib.mphip_conv3d_bwd_weight_supported(n, ci, co, d, h, w, k, prec)):
        prec = 0  # the exact fp32 kernel covers every shape
    ws_bytes = lib.mphip_conv3d_bwd_weight_workspace_bytes(n, ci, co, d, h, w, k, prec)
    if ws_bytes == 0:
        raise RuntimeError(f"conv3d_bwd_weight: unsupported shape {tuple(x.shape)} k={k}")
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, k, k, k), dtype=torch.float32, device=x.device)
    xr = (x_range if x_range is not None else tensor_range(x)) if prec == 1 else None   # None: the library measures x itself
    if roi is not None:   # dy is zero outside these per-frame boxes: the voxel tiles outside them are skipped
        _lib.check(lib.mphip_conv3d_bwd_weight_roi(_ptr(x), _ptr(xr), _ptr(dy), _ptr(dy_scale), _ptr(dw), _ptr(roi), n, ci, co, d, h, w, k, prec,
                                                   _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_bwd_weight_roi")
        return dw
    _lib.check(lib.mphip_conv3d_bwd_weight(_ptr(x), _ptr(xr), _ptr(dy), _ptr(dy_scale), _ptr(dw), n, ci, co, d, h, w, k, prec,
                                           _ptr(ws), ws_bytes, _stream()), "mphip_conv3d_bwd_weight")
    return dw


def groupnorm_bwd(x, y, dy, stats, gamma, groups: int, relu, want_res: bool, beta=None, w2=None):
    """Backward of y = act(GroupNorm(x)*gamma+beta [*w2+b2] (+res)) -> (dx, dgamma, dbeta, dres|None[, dw2, db2]).
    `y` is the forward output (activation mask); stats = the forward (mean, rstd); relu: False/True or act code 0/1/2
    (2 = tanh(ReLU(.)))."""
    x, dy = _req(x, "x"), _req(dy, "dy")
    n, c = x.shape[0], x.shape[1]
    s = x.numel() // (n * c)
    act = int(relu)
    lib = _lib.load()
    gamma = _req(gamma.detach(), "gamma")
    dev = x.device
    if w2 is not None:
        w2 = _req(w2.detach(), "w2").reshape(-1)
        beta = _req(beta.detach(), "beta")
    ws_bytes = lib.mphip_groupnorm_bwd_workspace_bytes(n, c, s)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    dw2 = torch.empty(c, dtype=torch.float32, device=dev) if w2 is not None else None
    db2 = torch.empty(c, dtype=torch.float32, device=dev) if w2 is not None else None
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    if _os.environ.get("MPHIP_GN_BWD_FUSED", "1") != "0":   # reduce + apply (the fold is re-derived inside the apply): two launches, same bits
        _lib.check(lib.mphip_groupnorm_bwd(_ptr(x), _ptr(y), _ptr(dy), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(w2), _ptr(dx), _ptr(dres),
                                           _ptr(dgamma), _ptr(dbeta), _ptr(dw2), _ptr(db2), n, c, s, groups, act, _ptr(ws), ws_bytes, _stream()),
                   "mphip_groupnorm_bwd")
    else:                                                    # (dev A/B and the bitwise reference of the fused entry: reduce + fold, then apply)
        ab = torch.empty((n * groups, 2), dtype=torch.float32, device=dev)
        _lib.check(lib.mphip_groupnorm_bwd_reduce(_ptr(x), _ptr(y), _ptr(dy), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(w2),
                                                  _ptr(dgamma), _ptr(dbeta), _ptr(dw2), _ptr(db2), _ptr(ab), n, c, s, groups, act,
                                                  _ptr(ws), ws_bytes, _stream()), "mphip_groupnorm_bwd_reduce")
        _lib.check(lib.mphip_groupnorm_bwd_apply(_ptr(x), _ptr(y), _ptr(dy), _ptr(stats), _ptr(gamma), _ptr(w2), _ptr(ab), _ptr(dx),
                                                 _ptr(dres), n, c, s, groups, act, _stream()), "mphip_groupnorm_bwd_apply")
    if w2 is not None:
        return dx, dgamma, dbeta, dres, dw2, db2
    return dx, dgamma, dbeta, dres


def avgpool2_bwd(dout: torch.Tensor) -> torch.Tensor:
    dout = _req(dout, "dout")
    n, c, d, h, w = dout.shape
    dx = torch.empty((n, c, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.load().mphip_avgpool2_bwd(_ptr(dout), _ptr(dx), n * c, 2 * d, 2 * h, 2 * w, _stream()), "mphip_avgpool2_bwd")
    return dx


def upsample_trilinear2_bwd(dout: torch.Tensor) -> torch.Tensor:
    dout = _req(dout, "dout")
    n, c, d, h, w = dout.shape
    if d % 2 or h % 2 or w % 2:
        raise RuntimeError("upsample_trilinear2_bwd: gradient dims must be even")
    dx = torch.empty((n, c, d // 2, h // 2, w // 2), dtype=torch.float32, device=dout.device)
    lib = _lib.load()
    ws_bytes = lib.mphip_upsample_trilinear2_bwd_workspace_bytes(n * c, d // 2, h // 2, w // 2)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dout.device)
    _lib.check(lib.mphip_upsample_trilinear2_bwd(_ptr(dout), _ptr(dx), n * c, d // 2, h // 2, w // 2, _ptr(ws), ws_bytes, _stream()),
               "mphip_upsample_trilinear2_bwd")
    return dx


# ------------------------------------------------------------------ K10  backward of the warps / field composition
def warp_volume_bwd(v: torch.Tensor, field: torch.Tensor, dout: torch.Tensor, dsum: bool, want_v: bool = True,
                    want_field: bool = True):
    """(dv, dfield) of warp_volume (dsum=False) / warp_volume_dsum (dsum=True, dout [B,C,H,W])."""
    v, field, dout = _req(v, "v"), _req(field, "warp_field"), _req(dout, "dout")
    b, c, d, h, w = v.shape
    want = (b, c, h, w) if dsum else (b, c, d, h, w)
    if tuple(dout.shape) != want:
        raise RuntimeError(f"warp_volume_bwd: dout {tuple(dout.shape)} != {want}")
    dev = v.device
    lib = _lib.load()
    dv = torch.empty_like(v) if want_v else None
    dfield = torch.empty_like(field) if want_field else None
    ws_bytes = lib.mphip_warp_volume_bwd_workspace_bytes(b, c, d, h, w)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    _lib.check(lib.mphip_warp_volume_bwd(_ptr(v), _ptr(field), _ptr(linspace_table(d, dev)), _ptr(linspace_table(h, dev)),
                                         _ptr(linspace_table(w, dev)), _ptr(dout), _ptr(dv), _ptr(dfield), b, c, d, h, w,
                                         field.shape[2], field.shape[3], field.shape[4], int(dsum), _ptr(ws), ws_bytes,
                                         _stream()), "mphip_warp_volume_bwd")
    return dv, dfield


def warp_field_compose_bwd(dw: torch.Tensor, em_shape, want_theta: bool = True, want_em: bool = True):
    """(dtheta [B,3,4], dem [B,3,eD,eH,eW]) of warp_field_compose given dw [B,3,G,G,G]."""
    dw = _req(dw, "dw")
    b, _, g = dw.shape[0], dw.shape[1], dw.shape[2]
    lib = _lib.load()
    dtheta = torch.empty((b, 3, 4), dtype=torch.float32, device=dw.device) if want_theta else None
    dem = torch.empty((b, 3) + tuple(em_shape[2:]), dtype=torch.float32, device=dw.device) if want_em else None
    ws_bytes = lib.mphip_warp_field_compose_bwd_workspace_bytes(b, g)
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=dw.device)
    _lib.check(lib.mphip_warp_field_compose_bwd(_ptr(dw), _ptr(affine_base_table(g, dw.device)), _ptr(dtheta), _ptr(dem), b,
                                                em_shape[2], em_shape[3], em_shape[4], g, _ptr(ws), ws_bytes, _stream()),
               "mphip_warp_field_compose_bwd")
    return dtheta, dem


def rt_theta_bwd(rotation_deg: torch.Tensor, translation: torch.Tensor, dtheta: torch.Tensor, invert: bool):
    rotation_deg, translation, dtheta = _req(rotation_deg, "rotation"), _req(translation, "translation"), _req(dtheta, "dtheta")
    b = rotation_deg.shape[0]
    drot, dtr = torch.empty_like(rotation_deg), torch.empty_like(translation)
    _lib.check(_lib.load().mphip_rt_theta_bwd(_ptr(rotation_deg), _ptr(translation), _ptr(dtheta), _ptr(drot), _ptr(dtr), b,
                                              int(bool(invert)), _stream()), "mphip_rt_theta_bwd")
    return drot, dtr


def upsample_nearest_bwd(dout: torch.Tensor, scale: Tuple[int, int, int]) -> torch.Tensor:
    dout = _req(dout, "dout")
    n, c, d, h, w = dout.shape
    sd, sh, sw = (int(v) for v in scale)
    if d % sd or h % sh or w % sw:
        raise RuntimeError("upsample_nearest_bwd: gradient dims are not multiples of the scale factors")
    dx = torch.empty((n, c, d // sd, h // sh, w // sw), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.load().mphip_upsample_nearest_bwd(_ptr(dout), _ptr(dx), n * c, d // sd, h // sh, w // sw, sd, sh, sw, _stream()),
               "mphip_upsample_nearest_bwd")
    return dx


def small_gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, a2: Optional[torch.Tensor] = None,
               bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """op(a (+a2)) @ op(b) (+bias) for the generators' tiny dense heads and their gradients (2-D contiguous fp32)."""
    a, b = _req(a, "a"), _req(b, "b")
    if a2 is not None:
        a2 = _req(a2, "a2")
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    k2, n = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if k != k2:
        raise RuntimeError(f"small_gemm: inner dims differ ({k} vs {k2})")
    sam, sak = (1, a.shape[1]) if trans_a else (a.shape[1], 1)
    sbk, sbn = (1, b.shape[1]) if trans_b else (b.shape[1], 1)
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().mphip_small_gemm(_ptr(a), _ptr(a2), _ptr(b), _ptr(bias), _ptr(out), m, n, k, sam, sak, sbk, sbn,
                                            _stream()), "mphip_small_gemm")
    return out
