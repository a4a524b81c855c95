"""ctypes wrapper of the plain-C oracle (oracle/hotpath_c.c -> libmphip_oracle.so).

TEST INFRASTRUCTURE — NOT PRODUCT CODE (see the header of hotpath_c.c).  Takes/returns torch CPU
fp32 tensors.  Composes the C routines into the same graph as oracle/hotpath_ref.py so the two
restatements (ATen-based and ATen-free) can be checked against each other and against the
golden fixtures.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmphip_oracle.so")
        if not os.path.isfile(path):
            import subprocess

            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_TABLES = None


def _tables():
    """The reference host's torch.linspace tables captured as data by make_golden.capture_tables
    (same JSON the product reads; data, not code).  Sizes not captured use the local torch."""
    global _TABLES
    if _TABLES is None:
        import json

        import numpy as np

        path = os.path.join(os.path.dirname(_HERE), "megaportrait-hack_amd", "data", "linspace_tables.json")
        with open(path) as f:
            raw = json.load(f)
        _TABLES = {kind: {int(n): torch.from_numpy(np.array(b, dtype=np.uint32).view(np.float32).copy())
                          for n, b in raw[kind].items()} for kind in ("linspace", "affine_base")}
    return _TABLES


def linspace(n):
    t = _tables()["linspace"].get(n)
    return (t if t is not None else torch.linspace(-1, 1, n)).contiguous()


def affine_base(g):
    t = _tables()["affine_base"].get(g)
    return (t if t is not None else torch.linspace(-1, 1, g) * (g - 1) / g).contiguous()


def _c(t):
    assert t.dtype == torch.float32 and not t.is_cuda
    return t.contiguous()


def affine_grid3d(theta, G):
    theta = _c(theta)
    B = theta.shape[0]
    base = affine_base(G)
    out = torch.empty(B, 3, G, G, G)
    lib().orc_affine_grid3d(_p(theta), _p(base), B, G, _p(out))
    return out


def resize_trilinear(x, size, align_corners):
    x = _c(x)
    n, c, d, h, w = x.shape
    out = torch.empty(n, c, *size)
    lib().orc_resize_trilinear(_p(x), _p(out), n * c, d, h, w, size[0], size[1], size[2], int(align_corners))
    return out


def upsample_nearest(x, scale):
    x = _c(x)
    n, c, d, h, w = x.shape
    out = torch.empty(n, c, d * scale[0], h * scale[1], w * scale[2])
    lib().orc_upsample_nearest(_p(x), _p(out), n * c, d, h, w, *scale)
    return out


def avgpool2(x):
    x = _c(x)
    n, c, d, h, w = x.shape
    out = torch.empty(n, c, d // 2, h // 2, w // 2)
    lib().orc_avgpool2(_p(x), _p(out), n * c, d, h, w)
    return out


def groupnorm(x, groups, gamma, beta, w2=None, b2=None, residual=None, relu=False, eps=1e-5):
    x = _c(x)
    n, c = x.shape[:2]
    s = x.numel() // (n * c)
    out = torch.empty_like(x)
    w2c = None if w2 is None else _c(w2.reshape(-1))
    b2c = None if b2 is None else _c(b2.reshape(-1))
    resc = None if residual is None else _c(residual)
    lib().orc_groupnorm(_p(x), _p(out), n, c, s, groups, _p(_c(gamma)), _p(_c(beta)), _p(w2c), _p(b2c), _p(resc),
                        int(relu), ctypes.c_float(eps))
    return out


def conv3d(x, weight, bias):
    x, weight = _c(x), _c(weight)
    n, ci, d, h, w = x.shape
    co, k = weight.shape[0], weight.shape[2]
    out = torch.empty(n, co, d, h, w)
    lib().orc_conv3d(_p(x), _p(weight), _p(None if bias is None else _c(bias)), _p(out), n, ci, co, d, h, w, k)
    return out


def warp_coords(warp_field, D, H, W):
    """-> (coords [B,D,H,W,3] fp32, idx int32): the bit-exact index contract."""
    f = resize_trilinear(warp_field, (D, H, W), True)
    B = f.shape[0]
    coords = torch.empty(B, D, H, W, 3)
    idx = torch.empty(B, D, H, W, 3, dtype=torch.int32)
    ld, lh, lw = (linspace(n) for n in (D, H, W))
    lib().orc_warp_coords(_p(f), _p(ld), _p(lh), _p(lw), B, D, H, W, _p(coords), _p(idx))
    return coords, idx


def apply_warping_field(v, warp_field, dsum=False):
    v = _c(v)
    B, C, D, H, W = v.shape
    coords, _ = warp_coords(warp_field, D, H, W)
    out = torch.empty(B, C, H, W) if dsum else torch.empty_like(v)
    lib().orc_grid_sample3d(_p(v), _p(coords), B, C, D, H, W, _p(out), int(dsum))
    return out


def matmul(a, m, bias=None, trans=False):
    a, m = _c(a), _c(m)
    B, K = a.shape
    N = m.shape[0] if trans else m.shape[1]
    out = torch.empty(B, N)
    lib().orc_matmul(_p(a), _p(m), _p(None if bias is None else _c(bias)), _p(out), B, K, N, int(trans))
    return out


def relu_tanh(x):
    x = _c(x)
    out = torch.empty_like(x)
    lib().orc_relu_tanh(_p(x), _p(out), ctypes.c_size_t(x.numel()))
    return out


# ---- graph composition (same structure as hotpath_ref.py; model.py line cites there) ----------
def resblock3d_adaptive(x, sd, p):
    def agn(t, q, residual=None):
        return groupnorm(t, 32, sd[q + "group_norm.weight"], sd[q + "group_norm.bias"], sd[q + "weight"], sd[q + "bias"],
                         residual=residual, relu=True)

    out = agn(conv3d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"]), p + "norm1.")
    out = conv3d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    res = conv3d(x, sd[p + "residual_conv.weight"], sd[p + "residual_conv.bias"]) if (p + "residual_conv.weight") in sd else x
    return agn(out, p + "norm2.", residual=res)


def flowfield(s, sd, p):
    b = s.shape[0]
    x = matmul(s, sd[p + "conv1x1.weight"].reshape(2048, 512), sd[p + "conv1x1.bias"], trans=True).view(b, 512, 4, 1, 1)
    for k, up in enumerate(((2, 2, 2), (2, 2, 2), (1, 2, 2), (1, 2, 2)), start=1):
        x = upsample_nearest(resblock3d_adaptive(x, sd, f"{p}resblock{k}."), up)
    x = conv3d(x, sd[p + "conv3x3x3.weight"], sd[p + "conv3x3x3.bias"])
    return relu_tanh(groupnorm(x, 1, sd[p + "gn.weight"], sd[p + "gn.bias"]))


def warp_generator(R, t, z, e, sd, p, invert):
    from . import hotpath_ref

    s = matmul(z + e, sd[p + "adaptive_matrix_gamma"])
    em = flowfield(s, sd, p + "flowfield.")
    theta = hotpath_ref.affine_theta(R, t, invert)  # 4x4 assembly/inverse: torch (LAPACK) as in the reference
    return affine_grid3d(theta, 64) + resize_trilinear(em, (64, 64, 64), False)


def resblock3d(x, sd, p):
    idn = conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"]) if (p + "shortcut.weight") in sd else x
    out = groupnorm(conv3d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"]), 32, sd[p + "gn1.weight"], sd[p + "gn1.bias"],
                    relu=True)
    out = conv3d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    return groupnorm(out, 32, sd[p + "gn2.weight"], sd[p + "gn2.bias"], residual=idn, relu=True)


def g3d(x, sd, p="G3d."):
    x = avgpool2(resblock3d(x, sd, p + "downsampling.0."))
    x = avgpool2(resblock3d(x, sd, p + "downsampling.2."))
    x = avgpool2(resblock3d(x, sd, p + "downsampling.4."))
    x = resblock3d(x, sd, p + "downsampling.6.")
    for name in ("upsampling.0.", "upsampling.2.", "upsampling.4."):
        x = resblock3d(x, sd, p + name)
        n, c, d, h, w = x.shape
        x = resize_trilinear(x, (2 * d, 2 * h, 2 * w), True)
    return conv3d(x, sd[p + "final_conv.weight"], sd[p + "final_conv.bias"])


def hot_slice(vs, es, Rs, ts, zs, Rd, td, zd, sd):
    w_s2c = warp_generator(Rs, ts, zs, es, sd, "warp_generator_s2c.", True)
    vc2d = g3d(apply_warping_field(vs, w_s2c), sd)
    w_c2d = warp_generator(Rd, td, zd, es, sd, "warp_generator_c2d.", False)
    return apply_warping_field(vc2d, w_c2d, dsum=True)
