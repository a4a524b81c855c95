"""ctypes binding of libmphip.so (C ABI declared in include/mphip.h).

This is the only way the Python host reaches the HIP kernels: device pointers come from
`torch.Tensor.data_ptr()`, the stream from `torch.cuda.current_stream()`.  There is NO CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPHIP_LIB", os.path.join(_HERE, "libmphip.so"))  # MPHIP_LIB: dev override (ablation builds)
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

_c_float_p = ctypes.c_void_p  # device pointers travel as raw addresses
_i = ctypes.c_int
_sz = ctypes.c_size_t
_p = ctypes.c_void_p

class PackJob(ctypes.Structure):
    """mphip_pack_job (include/mphip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("like", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int),
                ("k", ctypes.c_int), ("precision", ctypes.c_int), ("transposed", ctypes.c_int)]


# name -> (restype, argtypes); mirrors include/mphip.h one to one.
SIGNATURES = {
    "mphip_version": (_i, []),
    "mphip_last_error": (ctypes.c_char_p, []),
    "mphip_rt_theta": (_i, [_p, _p, _p, _i, _i, _p]),
    "mphip_warp_field_compose": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mphip_warp_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mphip_warp_corner_image_bytes": (_sz, [_i, _i]),
    "mphip_warp_corner_image": (_i, [_p, _p, _sz] + [_i] * 5 + [_p]),
    "mphip_warp_volume_coords_img": (_i, [_p] * 4 + [_i] * 5 + [_p, _sz, _p, _p]),
    "mphip_warp_volume": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_warp_volume_dsum": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_warp_volume_dsum_shared": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_conv3d_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "mphip_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "mphip_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_conv3d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "mphip_absmax_range": (_i, [_p, _sz, _p, _p]),
    "mphip_conv3d_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_conv3d_gn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "mphip_conv3d_gn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _p, _sz, _p]),
    "mphip_conv3d_gn_table_fwd": (_i, [_p] * 12 + [_i] * 9 + [ctypes.c_float, _p, _sz, _p]),
    "mphip_conv3d_time_next_launch": (None, [_p, _p]),
    "mphip_groupnorm_affine_table": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mphip_conv3d_gnin_gn_fwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _p] + [_i] * 9 + [ctypes.c_float, _p, _sz, _p]),
    "mphip_conv3d_gnin_fwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_conv3d_splits": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "mphip_conv3d_fwd_split": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_groupnorm_stats_split": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, ctypes.c_float, _p]),
    "mphip_groupnorm_apply_split": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p] + [_i] * 12 + [_p]),
    "mphip_groupnorm_small_fused": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p] + [_i] * 6 + [ctypes.c_float] + [_i] * 5 + [_p]),
    "mphip_groupnorm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mphip_groupnorm_stats": (_i, [_p, _p, _i, _i, _i, _i, ctypes.c_float, _p, _sz, _p]),
    "mphip_groupnorm_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mphip_avgpool2": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_upsample_trilinear2": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_upsample_nearest": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mphip_upsample_trilinear": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mphip_upsample_trilinear_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mphip_add_matmul": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mphip_grad_prep_workspace_bytes": (_sz, [_i, _i, _i]),
    "mphip_grad_prep": (_i, [_p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "mphip_pack_conv_weight_bwd_data": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_pack_conv_weight_bwd_data_like": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "mphip_conv3d_bwd_data": (_i, [_p, _p, _p, _p] + [_i] * 8 + [_p, _sz, _p]),
    "mphip_conv3d_bwd_weight_supported": (_i, [_i] * 8),
    "mphip_conv3d_bwd_weight_workspace_bytes": (_sz, [_i] * 8),
    "mphip_conv3d_bwd_weight": (_i, [_p, _p, _p, _p, _p] + [_i] * 8 + [_p, _sz, _p]),
    "mphip_groupnorm_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "mphip_groupnorm_bwd_reduce": (_i, [_p] * 12 + [_i] * 5 + [_p, _sz, _p]),
    "mphip_groupnorm_bwd_apply": (_i, [_p] * 9 + [_i] * 5 + [_p]),
    "mphip_groupnorm_bwd": (_i, [_p] * 13 + [_i] * 5 + [_p, _sz, _p]),
    "mphip_upsample_nearest_bwd": (_i, [_p, _p] + [_i] * 7 + [_p]),
    "mphip_small_gemm": (_i, [_p] * 5 + [_i] * 3 + [ctypes.c_long] * 4 + [_p]),
    "mphip_warp_coords": (_i, [_p] * 5 + [_i] * 7 + [_p]),
    "mphip_warp_volume_bwd_workspace_bytes": (_sz, [_i] * 5),
    "mphip_warp_volume_bwd": (_i, [_p] * 8 + [_i] * 9 + [_p, _sz, _p]),
    "mphip_warp_field_compose_bwd_workspace_bytes": (_sz, [_i, _i]),
    "mphip_warp_field_compose_bwd": (_i, [_p] * 4 + [_i] * 5 + [_p, _sz, _p]),
    "mphip_rt_theta_bwd": (_i, [_p] * 5 + [_i, _i, _p]),
    "mphip_f16x3_saturation_count": (_i, [_p, _i]),
    "mphip_conv3d_set_half_products": (_i, [_i]),
    "mphip_avgpool2_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_upsample_trilinear2_bwd_workspace_bytes": (_sz, [_i] * 4),
    "mphip_upsample_trilinear2_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "mphip_warp_field_coords": (_i, [_p] * 7 + [_i] * 6 + [_p]),
    "mphip_flowfield_conv_gn_supported": (_i, [_i] * 7),
    "mphip_flowfield_out_workspace_bytes": (_sz, [_i]),
    "mphip_flowfield_out": (_i, [_p] * 6 + [_i, ctypes.c_float, _p, _sz, _p]),
    "mphip_flowfield_conv_gn": (_i, [_p] * 11 + [_i] * 11 + [ctypes.c_float, _i, _p]),
    "mphip_flowfield_compact_weight_bytes": (_sz, [_i] * 5),
    "mphip_flowfield_compact_weight": (_i, [_p, _p, _i, _i, _p]),
    "mphip_flowfield_conv_gn_compact": (_i, [_p] * 11 + [_i] * 11 + [ctypes.c_float, _i, _p]),
    "mphip_warp_volume_coords": (_i, [_p] * 4 + [_i] * 5 + [_p, _sz, _p]),
    "mphip_warp_sample_box": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mphip_conv3d_roi_granule": (_i, [_i] * 8 + [_p]),
    "mphip_conv3d_roi_workspace_bytes": (_sz, [_i] * 8),
    "mphip_conv3d_fwd_roi": (_i, [_p] * 6 + [_i] * 9 + [_p, _sz, _p]),
    "mphip_conv3d_bwd_data_roi": (_i, [_p] * 5 + [_i] * 8 + [_p, _sz, _p]),
    "mphip_conv3d_bwd_weight_roi": (_i, [_p] * 6 + [_i] * 8 + [_p, _sz, _p]),
    "mphip_upsample_trilinear2_roi": (_i, [_p, _p, _p] + [_i] * 9 + [_p]),
    "mphip_warp_volume_dsum_coords": (_i, [_p, _p, _p] + [_i] * 6 + [_p]),
    "mphip_hot_slice_plan_create": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mphip_hot_slice_plan_set_tables": (_i, [_p, _p, _p, _p, _p]),
    "mphip_hot_slice_plan_refresh": (_i, [_p, _p, _p, _i]),
    "mphip_hot_slice_plan_profile": (_i, [_p, _i]),
    "mphip_hot_slice_plan_profile_read": (_i, [_p, _i, _p, _p]),
    "mphip_hot_slice_plan_set_precision": (_i, [_p, _i]),
    "mphip_hot_slice_workspace_bytes": (_sz, [_p, _i]),
    "mphip_hot_slice_forward": (_i, [_p] * 10 + [_i, _p, _sz, _p]),
    "mphip_g3d_workspace_bytes": (_sz, [_p, _i]),
    "mphip_g3d_forward": (_i, [_p, _p, _p, _p, _i, _p, _sz, _p]),
    "mphip_hot_slice_plan_destroy": (None, [_p]),
    "mphip_debug_mfma_sol": (_i, [_p, _i, _i, _i, _p]),
    "mphip_conv3d_kernel_variant": (_i, [_i] * 8),
    "mphip_build_flags": (_i, []),
    "mphip_graph_memsets_to_kernels": (_i, [_p, ctypes.POINTER(ctypes.c_int)]),
    "mphip_graph_memset_nodes_left": (_i, [_p, ctypes.POINTER(ctypes.c_int)]),
    "mphip_pack_table_create": (_i, [_p, _i, ctypes.POINTER(ctypes.c_void_p)]),
    "mphip_pack_table_run": (_i, [_p, _p]),
    "mphip_pack_table_destroy": (_i, [_p]),
    "mphip_debug_dma_stream": (_i, [_p, _i, _i, _i, _i, _p, _i, _p]),
}

_lib = None

# The ABI version the SIGNATURES table above mirrors.  Checked against the library at load time, and against include/mphip.h by
# tests/test_host.py — NOT read from the header at run time: a relocated / installed package ships libmphip.so without the repository's
# include/ directory.
EXPECTED_ABI_VERSION = 14


def header_abi_version() -> int:
    """MPHIP_ABI_VERSION of include/mphip.h when the header is there (a source checkout), else the version this binding was written
    for (EXPECTED_ABI_VERSION)."""
    import re

    path = os.path.join(os.path.dirname(_HERE), "include", "mphip.h")
    if not os.path.isfile(path):
        return EXPECTED_ABI_VERSION
    with open(path) as f:
        m = re.search(r"#define\s+MPHIP_ABI_VERSION\s+(\d+)", f.read())
    if m is None:
        raise RuntimeError("include/mphip.h does not define MPHIP_ABI_VERSION")
    return int(m.group(1))


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libmphip.so (hipcc cross-compiles without a GPU)."""
    if force or not os.path.isfile(LIB_PATH) or _stale():
        subprocess.run(["bash", BUILD_SCRIPT, LIB_PATH], check=True)
    return LIB_PATH


def _stale() -> bool:
    src_dir = os.path.join(_HERE, "csrc")
    lib_m = os.path.getmtime(LIB_PATH)
    inc = os.path.join(os.path.dirname(_HERE), "include", "mphip.h")
    files = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".sh"))] + [inc]
    return any(os.path.getmtime(f) > lib_m for f in files if os.path.isfile(f))


def load() -> ctypes.CDLL:
    """Loads libmphip.so; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    want, got = header_abi_version(), lib.mphip_version()
    if want != got:   # a stale build next to a newer header (or the reverse): arguments would be passed shifted
        raise RuntimeError(f"{LIB_PATH} was built with MPHIP_ABI_VERSION {got} but include/mphip.h declares {want}; rebuild "
                           "it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    if lib.mphip_build_flags() & 1 and os.environ.get("MPHIP_ALLOW_ABLATED") != "1":
        raise RuntimeError(f"{LIB_PATH} is a development variant built with a timing-only ablation (csrc/mphip_ablate.h): its kernels "
                           "compute wrong results by design.  Set MPHIP_ALLOW_ABLATED=1 to load it for a measurement.")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mphip_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
