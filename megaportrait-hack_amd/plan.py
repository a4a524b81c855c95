"""One-call entries of libmphip.so for the hot slice and G3d (include/mphip.h "one-call entries", csrc/plan.hip).

`HotSlicePlan` wraps `mphip_hot_slice_plan_*`: it is built from a module that carries the reference's attribute names
(`warp_generator_s2c`, `warp_generator_c2d`, `G3d` — GbaseHotSlice, gbase.Gbase, or the reference's own Gbase), hands the
C side the state-dict tensors by name, and from then on ONE ctypes call per step replaces the ~135 per-op calls of the
Python schedule (model._HotSliceRunner._run): the launches, their order and the results are identical; what changes is
the host cost per step (a single frame is launch-bound).  Inference only.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib, ops

_P = ctypes.c_void_p
_PREFIXES = ("warp_generator_s2c.", "warp_generator_c2d.", "G3d.")


def _hot_tensors(module: torch.nn.Module, g3d_only: bool) -> Dict[str, torch.Tensor]:
    """state-dict key -> parameter tensor (the live storage, not a copy), for the keys the plan reads."""
    out = {}
    for name, p in module.named_parameters():
        if not name.startswith(_PREFIXES) or "adaptive_matrix_beta" in name:
            continue
        if g3d_only and not name.startswith("G3d."):
            continue
        t = p.detach()
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"HotSlicePlan: parameter {name} must be a float32 CUDA tensor (got {t.dtype} on {t.device})")
        if not t.is_contiguous():
            raise RuntimeError(f"HotSlicePlan: parameter {name} is not contiguous")
        out[name] = t
    return out


class HotSlicePlan:
    def __init__(self, module: torch.nn.Module, dims=(96, 16, 64, 64), g3d_only: bool = False, single_stream: bool = False,
                 full_final_conv: bool = False):
        """full_final_conv: evaluate G3d's last upsample + conv on every voxel.  Default: demand-driven — only the tiles the
        final warp (apply_warping_field + depth sum) reads are produced (include/mphip.h); same output bits either way."""
        self.lib = _lib.load()
        self.module = module
        self.dims = tuple(int(v) for v in dims)
        self.g3d_only = bool(g3d_only)
        self._handle = _P()
        tensors = _hot_tensors(module, self.g3d_only)
        if not tensors:
            raise RuntimeError("HotSlicePlan: the module has no warp_generator_s2c / warp_generator_c2d / G3d parameters")
        self.device = next(iter(tensors.values())).device
        names, ptrs = self._tables(tensors)
        flags = (1 if g3d_only else 0) | (2 if single_stream else 0) | (4 if full_final_conv else 0)
        c, d, h, w = self.dims
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mphip_hot_slice_plan_create(names, ptrs, len(tensors), c, d, h, w, flags, ctypes.byref(self._handle)),
                       "mphip_hot_slice_plan_create")
        # the index pipeline's host-built tables: always the binding's own (captured bits for 16 / 64, this host's torch.linspace
        # otherwise), so both paths use the same tables at every size
        self._tbl = (ops.linspace_table(d, self.device), ops.linspace_table(h, self.device), ops.linspace_table(w, self.device),
                     ops.affine_base_table(64, self.device))
        _lib.check(self.lib.mphip_hot_slice_plan_set_tables(self._handle, *(_P(t.data_ptr()) for t in self._tbl)),
                   "mphip_hot_slice_plan_set_tables")
        self._key = self._weights_key(tensors)
        self._ws: Dict[tuple, torch.Tensor] = {}

    def _tables(self, tensors):
        self._keep = tensors   # the plan holds raw pointers: keep the tensors alive
        live = dict(self.module.named_parameters())
        self._params = [live[k] for k in tensors]   # same order as the key below
        names = (ctypes.c_char_p * len(tensors))(*[k.encode() for k in tensors])
        ptrs = (_P * len(tensors))(*[t.data_ptr() for t in tensors.values()])
        return names, ptrs

    @staticmethod
    def _weights_key(tensors):
        return (ops.weight_epoch(),) + tuple((t.data_ptr(), t._version) for t in tensors.values())

    def _sync_precision(self):
        prec = ops.get_conv_precision()   # ops.set_conv_precision / MPHIP_CONV_PRECISION apply to the plan's launches too
        if prec != getattr(self, "_prec", 1):
            _lib.check(self.lib.mphip_hot_slice_plan_set_precision(self._handle, prec), "mphip_hot_slice_plan_set_precision")
            self._prec = prec

    def _sync_weights(self):
        """Re-pack (on the forward's stream) when a parameter changed: in-place update, load_state_dict, .to(), or
        ops.invalidate_packs() — the same conditions under which model._PackCache rebuilds its packs.
        The per-forward check reads (data_ptr, _version) of the ~200 parameter tensors the plan bound (a B=1 step is 1.2 ms: walking
        named_parameters() and detaching every tensor each call cost more host time than the check needs — ADVICE r3); the module tree
        is walked again only when that fingerprint or the pack epoch moved.  Replacing a Parameter OBJECT of the module (rather than
        its contents) needs ops.invalidate_packs(), like every other cache keyed on the tensors."""
        self._sync_precision()
        fast = (ops.weight_epoch(),) + tuple((q.data_ptr(), q._version) for q in self._params)   # (the live Parameters: .to() shows as a new pointer)
        if fast == self._key and not ops.repacking():
            return
        tensors = _hot_tensors(self.module, self.g3d_only)
        key = self._weights_key(tensors)
        if key != self._key or ops.repacking():
            names, ptrs = self._tables(tensors)
            _lib.check(self.lib.mphip_hot_slice_plan_refresh(self._handle, names, ptrs, len(tensors)), "mphip_hot_slice_plan_refresh")
            self._key = key

    def _workspace(self, kind: str, b: int, nbytes: int) -> torch.Tensor:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        key = (kind, b, stream)   # two forwards in flight on different streams must not share a workspace
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def forward(self, vs, es, Rs, ts, zs, Rd, td, zd) -> torch.Tensor:
        if self.g3d_only:
            raise RuntimeError("HotSlicePlan(g3d_only=True) has no generators: use g3d()")
        c, d, h, w = self.dims
        ins = [ops._req(t, n) for t, n in ((vs, "vs"), (es, "es"), (Rs, "Rs"), (ts, "ts"), (zs, "zs"), (Rd, "Rd"), (td, "td"), (zd, "zd"))]
        b = ins[0].shape[0]
        if tuple(ins[0].shape) != (b, c, d, h, w):
            raise RuntimeError(f"HotSlicePlan.forward: vs {tuple(ins[0].shape)} does not match the plan's volume {(c, d, h, w)}")
        for t, n, k in zip(ins[1:], ("es", "Rs", "ts", "zs", "Rd", "td", "zd"), (512, 3, 3, 512, 3, 3, 512)):
            if t.numel() != b * k:
                raise RuntimeError(f"HotSlicePlan.forward: {n} has {t.numel()} elements, expected {b}x{k}")
        self._sync_weights()
        nbytes = self.lib.mphip_hot_slice_workspace_bytes(self._handle, b)
        ws = self._workspace("slice", b, nbytes)
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mphip_hot_slice_forward(self._handle, *(_P(t.data_ptr()) for t in ins), _P(out.data_ptr()), b, _P(ws.data_ptr()),
                                                    ws.numel(), ops._stream()), "mphip_hot_slice_forward")
        return out

    __call__ = forward

    def g3d(self, x: torch.Tensor, x_range: Optional[torch.Tensor] = None) -> torch.Tensor:
        c, d, h, w = self.dims
        x = ops._req(x, "x")
        b = x.shape[0]
        if tuple(x.shape) != (b, c, d, h, w):
            raise RuntimeError(f"HotSlicePlan.g3d: x {tuple(x.shape)} does not match the plan's volume {(c, d, h, w)}")
        self._sync_weights()
        rng = x_range if x_range is not None else ops.tensor_range(x)
        nbytes = self.lib.mphip_g3d_workspace_bytes(self._handle, b)
        ws = self._workspace("g3d", b, nbytes)
        y = torch.empty_like(x)
        _lib.check(self.lib.mphip_g3d_forward(self._handle, _P(x.data_ptr()), None if rng is None else _P(rng.data_ptr()), _P(y.data_ptr()), b,
                                              _P(ws.data_ptr()), ws.numel(), ops._stream()), "mphip_g3d_forward")
        return y

    def profile(self, enable: bool = True) -> None:
        """HIP events around every launch of the dominant conv (bench.py's `roofline`), on the stream it is launched on."""
        _lib.check(self.lib.mphip_hot_slice_plan_profile(self._handle, int(bool(enable))), "mphip_hot_slice_plan_profile")

    def profile_read(self, kind: int = 0):
        """(sum of launch durations in ms, launches) since the last read; kind 0 = full launches, 1 = the demand-driven one."""
        tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(self.lib.mphip_hot_slice_plan_profile_read(self._handle, kind, ctypes.byref(tot), ctypes.byref(cnt)),
                   "mphip_hot_slice_plan_profile_read")
        return float(tot.value), int(cnt.value)

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            with torch.cuda.device(self.device):
                self.lib.mphip_hot_slice_plan_destroy(self._handle)
            self._handle = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
