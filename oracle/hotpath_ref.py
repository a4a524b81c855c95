"""CPU restatement (torch fp32, functional) of the MegaPortraits Gbase hot slice.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this file.  The product path
(megaportrait-hack_amd/) never routes through it and has no CPU fallback.

What it restates: johndpope/MegaPortrait-hack `model.py:1151-1171` (the slice of
`Gbase.forward` between the 2D encoders and G2d) and every block it calls.  The
reference's arithmetic lives in PyTorch ATen (third party, un-vendored, version
unpinned by requirements.txt; pinned de facto by this image's torch 2.10.0), so
each function below re-derives the reference's *module graph* as a pure function
of a flat state-dict (the reference's own key names) and
calls the same ATen CPU primitives the reference would hit on a CPU host.  A
second, ATen-free restatement of the same ops in plain C (oracle/hotpath_c.c,
wrapped by oracle/hotpath_c.py) pins the bit-level index pipeline.

Pinning: tests/test_oracle.py checks every function here against
the imported reference (this container only) and tests/golden/*.npz holds the
outputs the reference produced for seeded inputs (generator:
oracle/make_golden.py), so the oracle stays pinned on the GPU box where
/root/reference does not exist.

All tensors fp32, NCDHW, CPU.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- a6
# Every conv of the restatement goes through CONV3D (default: ATen's conv3d).  Tests of the autocast policy (the reference's generator step
# runs under torch.cuda.amp.autocast(): conv3d on f16 operands, fp32 accumulate — train.py:145,188) swap in a conv that rounds its
# operands the way that policy does; the product never imports this module.
CONV3D = F.conv3d


def _conv3d(*args, **kwargs):
    return CONV3D(*args, **kwargs)


def conv3d_f16_operands(x, w, b=None, padding=0):
    """conv3d with input and weight rounded to float16 and the products accumulated exactly (float64): the arithmetic contract of an
    autocast(float16) conv3d, without any implementation's accumulation order."""
    y = F.conv3d(x.detach().half().double(), w.detach().half().double(), None if b is None else b.detach().double(), padding=padding)
    return y


def pow2_operand_scale(max_abs: float, top: int) -> float:
    """The power-of-two operand scale s of the split-f16 kernels: max|x| * s < 2^top (csrc/mphip_common.h scale_from_bits: top = 14 for
    activations / gradients; csrc/mphip_f16x3.h weight_scale: top = 15 for weights).  All-zero or non-finite tensors: 1."""
    if not (0.0 < max_abs < 3.0e38):
        return 1.0
    _, e = math.frexp(max_abs)          # max_abs = f * 2^e, f in [0.5, 1)
    return 2.0 ** (top - e)


def round_f16_at_scale(x: torch.Tensor, top: int = 14) -> torch.Tensor:
    """x rounded to float16 AT its power-of-two operand scale, returned unscaled in float64: the operand a one-product (autocast) launch
    of a direct-domain kernel multiplies (bwd-weight: x and dy).  Scaling by a power of two commutes with the rounding inside f16's
    normal range, so this equals x.half() except for values the unscaled cast would flush into f16 subnormals."""
    s = pow2_operand_scale(x.detach().abs().max().item(), top)
    return (x.detach().float() * s).half().double() / s


def conv3d_wino_f16_contract(x, w, b=None, padding=1):
    """The arithmetic contract of the F(2,3) conv kernels under the autocast policy (mphip_conv3d_set_half_products), stated without any
    implementation's accumulation order: k = 3, padding = 1, W even.  Along W the 3-tap filter runs in the Winograd F(2,3) domain; what
    is rounded to float16 is the TRANSFORMED operand pair, exactly as the kernels do it —
      input  (csrc/conv3d_f16x3_wino_pp.hip halo_write): t = Bt d on the scaled fp32 input (d0-d2, d1+d2, d2-d1, d1-d3, one fp32
             operation each), then rne_f16(t);
      filter (csrc/conv3d_f16x3.hip f16x3_pack_body): u = G g in double (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2), scaled, rounded
             double -> float -> f16;
    the products are accumulated exactly (float64), the output transform At = [[1,1,1,0],[0,1,-1,-1]] and the unscale are exact, then the
    bias.  A kernel that follows this contract differs from it only by its fp32 accumulation (~1e-6 of max|y|)."""
    assert padding == 1 and w.shape[2:] == (3, 3, 3) and x.shape[-1] % 2 == 0
    x, w = x.detach().float(), w.detach().float()
    W = x.shape[-1]
    sx = pow2_operand_scale(x.abs().max().item(), 14)
    sw = pow2_operand_scale(w.abs().max().item(), 15)
    xp = F.pad(x * sx, (1, 1))                                        # W halo; D / H are padded by the conv below
    d = [xp[..., i:i + W:2] for i in range(4)]                         # d_i of output pair j = x[2j - 1 + i]
    t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]          # fp32, one rounding each
    g = w.double()
    u = [g[..., 0], 0.5 * (g[..., 0] + g[..., 1] + g[..., 2]), 0.5 * (g[..., 0] - g[..., 1] + g[..., 2]), g[..., 2]]
    m = [F.conv3d(tp.half().double(), (up * sw).float().half().double().unsqueeze(-1), None, padding=(1, 1, 0)) for tp, up in zip(t, u)]
    y = torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], dim=-1).flatten(-2) / (sx * sw)
    return y if b is None else y + b.detach().double().view(1, -1, 1, 1, 1)


def rotation_matrix(rotation_deg: torch.Tensor) -> torch.Tensor:
    """model.py:811-856 — Euler degrees (alpha,beta,gamma)=(x,y,z) -> R = Rx @ (Ry @ Rz)."""
    r = rotation_deg * (torch.pi / 180.0)
    ca, sa = torch.cos(r[:, 0]), torch.sin(r[:, 0])
    cb, sb = torch.cos(r[:, 1]), torch.sin(r[:, 1])
    cg, sg = torch.cos(r[:, 2]), torch.sin(r[:, 2])
    zero, one = torch.zeros_like(ca), torch.ones_like(ca)

    def mat(rows):
        return torch.stack([torch.stack(row, dim=1) for row in rows], dim=1)

    rx = mat([[one, zero, zero], [zero, ca, -sa], [zero, sa, ca]])
    ry = mat([[cb, zero, sb], [zero, one, zero], [-sb, zero, cb]])
    rz = mat([[cg, -sg, zero], [sg, cg, zero], [zero, zero, one]])
    return torch.matmul(rx, torch.matmul(ry, rz))


def affine_theta(rotation_deg: torch.Tensor, translation: torch.Tensor, invert: bool) -> torch.Tensor:
    """model.py:790-804 — the [B,3,4] matrix handed to affine_grid."""
    b = rotation_deg.shape[0]
    a = torch.eye(4).repeat(b, 1, 1)
    a[:, :3, :3] = rotation_matrix(rotation_deg)
    a[:, :3, 3] = translation
    if invert:
        a = torch.inverse(a)
    return a[:, :3].contiguous()


def compute_rt_warp(rotation_deg, translation, invert=False, grid_size=64) -> torch.Tensor:
    """model.py:777-809 -> [B,3,G,G,G], channels (x,y,z)."""
    theta = affine_theta(rotation_deg, translation, invert)
    b = rotation_deg.shape[0]
    grid = F.affine_grid(theta, (b, 1, grid_size, grid_size, grid_size), align_corners=False)
    return grid.permute(0, 4, 1, 2, 3)


# --------------------------------------------------------------------------- a5
def adaptive_group_norm(x, sd: SD, prefix: str) -> torch.Tensor:
    """model.py:304-316 — GroupNorm(32,C) affine, then a second [1,C,1,1,1] affine."""
    n = F.group_norm(x, 32, sd[prefix + "group_norm.weight"], sd[prefix + "group_norm.bias"], 1e-5)
    return n * sd[prefix + "weight"] + sd[prefix + "bias"]


def resblock3d_adaptive(x, sd: SD, prefix: str) -> torch.Tensor:
    """model.py:385-408 (upsample flag never set on the hot path)."""
    out = _conv3d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1)
    out = F.relu(adaptive_group_norm(out, sd, prefix + "norm1."))
    out = _conv3d(out, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], padding=1)
    out = adaptive_group_norm(out, sd, prefix + "norm2.")
    if (prefix + "residual_conv.weight") in sd:
        res = _conv3d(x, sd[prefix + "residual_conv.weight"], sd[prefix + "residual_conv.bias"])
    else:
        res = x
    return F.relu(out + res)


# --------------------------------------------------------------------------- a4
_FLOW_UPS = ((2, 2, 2), (2, 2, 2), (1, 2, 2), (1, 2, 2))  # model.py:427-433


def flowfield(zs_sum: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """model.py:439-471.  zs_sum [B,512] (the reference passes [B,512,1,1]) -> [B,3,16,16,16]."""
    b = zs_sum.shape[0]
    x = F.conv2d(zs_sum.reshape(b, 512, 1, 1), sd[prefix + "conv1x1.weight"], sd[prefix + "conv1x1.bias"])
    x = x.view(-1, 512, 4, 1, 1)
    for k, up in enumerate(_FLOW_UPS, start=1):
        x = resblock3d_adaptive(x, sd, f"{prefix}resblock{k}.")
        x = F.interpolate(x, scale_factor=up, mode="nearest")  # nn.Upsample default mode
    x = _conv3d(x, sd[prefix + "conv3x3x3.weight"], sd[prefix + "conv3x3x3.bias"], padding=1)
    x = F.group_norm(x, 1, sd[prefix + "gn.weight"], sd[prefix + "gn.bias"], 1e-5)
    return torch.tanh(F.relu(x))


# ------------------------------------------------------------------------ a2/a3
def warp_generator(R, t, z, e, sd: SD, prefix: str, invert: bool, parts: bool = False):
    """model.py:938-975 (S2C: invert=True) / 989-1024 (C2D: invert=False) -> [B,3,64,64,64]."""
    assert R.shape == (z.shape[0], 3) and t.shape == (z.shape[0], 3) and z.shape == e.shape
    s = torch.matmul(z + e, sd[prefix + "adaptive_matrix_gamma"])  # right-multiply, no transpose
    em = flowfield(s, sd, prefix + "flowfield.")
    rt = compute_rt_warp(R, t, invert=invert, grid_size=64)
    em64 = F.interpolate(em, size=rt.shape[2:], mode="trilinear", align_corners=False)
    w = rt + em64
    if parts:
        return w, {"s": s, "em": em, "rt": rt, "em64": em64}
    return w


# --------------------------------------------------------------------------- a7
def resized_field(warp_field: torch.Tensor, D: int, H: int, W: int) -> torch.Tensor:
    """model.py:1036."""
    return F.interpolate(warp_field, size=(D, H, W), mode="trilinear", align_corners=True)


def normalized_grid(warp_field: torch.Tensor, D: int, H: int, W: int) -> torch.Tensor:
    """model.py:1036-1058 -> the [B,D,H,W,3] grid given to grid_sample."""
    b = warp_field.shape[0]
    f = resized_field(warp_field, D, H, W)
    d = torch.linspace(-1, 1, D)
    h = torch.linspace(-1, 1, H)
    w = torch.linspace(-1, 1, W)
    gd, gh, gw = torch.meshgrid(d, h, w, indexing="ij")
    grid = torch.stack((gw, gh, gd), dim=-1).unsqueeze(0).repeat(b, 1, 1, 1, 1)
    warped = grid + f.permute(0, 2, 3, 4, 1)
    return 2.0 * warped / torch.tensor([W - 1, H - 1, D - 1]) - 1.0


def apply_warping_field(v: torch.Tensor, warp_field: torch.Tensor) -> torch.Tensor:
    """model.py:1028-1065."""
    _, _, D, H, W = v.shape
    n = normalized_grid(warp_field, D, H, W)
    return F.grid_sample(v, n, mode="bilinear", padding_mode="border", align_corners=True)


def warp_coords(warp_field: torch.Tensor, D: int, H: int, W: int):
    """Clipped un-normalised sample coordinates and their floor indices, following
    ATen GridSampler.h:27-36,58-60 (align_corners=True, border): c=((n+1)/2)*(S-1);
    c=min(S-1,max(c,0)); i0=floor(c).  Returns (coords[B,D,H,W,3] fp32, idx int32) in (x,y,z) order."""
    n = normalized_grid(warp_field, D, H, W)
    size = torch.tensor([W - 1, H - 1, D - 1], dtype=torch.float32)
    c = ((n + 1.0) / 2.0) * size
    c = torch.minimum(size, torch.clamp_min(c, 0.0))
    return c, torch.floor(c).to(torch.int32)


def depth_projection(v: torch.Tensor) -> torch.Tensor:
    """model.py:1171 — a SUM over depth."""
    return torch.sum(v, dim=2)


# ----------------------------------------------------------------------- a8/a9
def resblock3d(x, sd: SD, prefix: str) -> torch.Tensor:
    """model.py:512-528."""
    if (prefix + "shortcut.weight") in sd:
        identity = _conv3d(x, sd[prefix + "shortcut.weight"], sd[prefix + "shortcut.bias"])
    else:
        identity = x
    out = _conv3d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1)
    out = F.relu(F.group_norm(out, 32, sd[prefix + "gn1.weight"], sd[prefix + "gn1.bias"], 1e-5))
    out = _conv3d(out, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], padding=1)
    out = F.group_norm(out, 32, sd[prefix + "gn2.weight"], sd[prefix + "gn2.bias"], 1e-5)
    return F.relu(out + identity)


def g3d(x, sd: SD, prefix: str = "G3d.") -> torch.Tensor:
    """model.py:571-597."""
    x = resblock3d(x, sd, prefix + "downsampling.0.")
    x = F.avg_pool3d(x, 2, 2)
    x = resblock3d(x, sd, prefix + "downsampling.2.")
    x = F.avg_pool3d(x, 2, 2)
    x = resblock3d(x, sd, prefix + "downsampling.4.")
    x = F.avg_pool3d(x, 2, 2)
    x = resblock3d(x, sd, prefix + "downsampling.6.")
    x = resblock3d(x, sd, prefix + "upsampling.0.")
    x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)
    x = resblock3d(x, sd, prefix + "upsampling.2.")
    x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)
    x = resblock3d(x, sd, prefix + "upsampling.4.")
    x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)
    return _conv3d(x, sd[prefix + "final_conv.weight"], sd[prefix + "final_conv.bias"], padding=1)


# --------------------------------------------------------------------------- f3 (next row)
def g2d_head(p: torch.Tensor, sd: SD, prefix: str = "G2d.") -> torch.Tensor:
    """model.py:718-719, 756-757: `reshape` Conv2d(96,1536,1) then `conv1x1` Conv2d(1536,512,1), no nonlinearity between."""
    x = F.conv2d(p, sd[prefix + "reshape.weight"], sd[prefix + "reshape.bias"])
    return F.conv2d(x, sd[prefix + "conv1x1.weight"], sd[prefix + "conv1x1.bias"])


def g2d_head_shapes() -> Dict[str, tuple]:
    return {"reshape.weight": (1536, 96, 1, 1), "reshape.bias": (1536,), "conv1x1.weight": (512, 1536, 1, 1), "conv1x1.bias": (512,)}


# --------------------------------------------------------------------------- f1 (next row)
_EAPP_TAIL_ORDER = ("resblock3D_96", "resblock3D_96_2", "resblock3D_96_1", "resblock3D_96_1_2", "resblock3D_96_2",
                    "resblock3D_96_2_2")   # model.py:276-290 — `resblock3D_96_2` is assigned twice (218,225) and applied twice


def eapp_tail3d(feat: torch.Tensor, sd: SD, prefix: str = "appearanceEncoder.") -> torch.Tensor:
    """Eapp's 3D tail, model.py:271-290: [B,1536,H,W] -> view [B,96,16,H,W] -> six applications of five
    ResBlock3D_Adaptive(96,96) blocks.  Accepts the 1536-channel map or an already reshaped volume."""
    vs = feat.view(feat.size(0), 96, 16, *feat.shape[2:]) if feat.dim() == 4 else feat
    for name in _EAPP_TAIL_ORDER:
        vs = resblock3d_adaptive(vs, sd, f"{prefix}{name}.")
    return vs


def eapp_tail_shapes() -> Dict[str, tuple]:
    sh = {}
    for name in sorted(set(_EAPP_TAIL_ORDER)):
        p = name + "."
        sh[p + "conv1.weight"] = (96, 96, 3, 3, 3)
        sh[p + "conv1.bias"] = (96,)
        sh[p + "conv2.weight"] = (96, 96, 3, 3, 3)
        sh[p + "conv2.bias"] = (96,)
        for n in ("norm1.", "norm2."):
            sh[p + n + "weight"] = (1, 96, 1, 1, 1)
            sh[p + n + "bias"] = (1, 96, 1, 1, 1)
            sh[p + n + "group_norm.weight"] = (96,)
            sh[p + n + "group_norm.bias"] = (96,)
    return sh


# --------------------------------------------------------------------------- a1
def hot_slice(vs, es, Rs, ts, zs, Rd, td, zd, sd: SD, stages: bool = False):
    """model.py:1151-1171: S2C field -> warp -> G3d -> C2D field -> warp -> sum over depth.
    `sd` uses Gbase's key names (warp_generator_s2c.*, warp_generator_c2d.*, G3d.*)."""
    w_s2c = warp_generator(Rs, ts, zs, es, sd, "warp_generator_s2c.", invert=True)
    vc = apply_warping_field(vs, w_s2c)
    vc2d = g3d(vc, sd, "G3d.")
    w_c2d = warp_generator(Rd, td, zd, es, sd, "warp_generator_c2d.", invert=False)
    vw = apply_warping_field(vc2d, w_c2d)
    proj = depth_projection(vw)
    if stages:
        return proj, {"w_s2c": w_s2c, "vc": vc, "vc2d": vc2d, "w_c2d": w_c2d}
    return proj


# ------------------------------------------------------------- deterministic data
def _lcg_uniform(n: int, seed: int) -> torch.Tensor:
    """Integer PRNG (64-bit LCG, top 24 bits) -> U[-1,1) fp32, reproducible anywhere.
    Vectorised with numpy uint64 wraparound (jump-ahead by exponentiation by squaring)."""
    import numpy as np

    a = np.uint64(6364136223846793005)
    c = np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        # state_i = A_i*s0 + C_i with (A_i, C_i) built by doubling
        A = np.empty(n, dtype=np.uint64)
        C = np.empty(n, dtype=np.uint64)
        A[0], C[0] = a, c
        filled = 1
        while filled < n:
            m = min(filled, n - filled)
            Ak, Ck = A[filled - 1], C[filled - 1]  # transform advancing by `filled` steps
            A[filled:filled + m] = A[:m] * Ak
            C[filled:filled + m] = A[:m] * Ck + C[:m]
            filled += m
        s0 = np.uint64(seed * 2654435761 + 88172645463325252 & 0xFFFFFFFFFFFFFFFF)
        st = A * s0 + C
    u = (st >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
    return torch.from_numpy((u * 2.0 - 1.0).astype(np.float32))


def seeded_tensor(shape, seed: int, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    n = int(math.prod(shape))
    return (_lcg_uniform(n, seed) * scale + shift).reshape(shape).contiguous()


def warp_generator_shapes() -> Dict[str, tuple]:
    """SURVEY.md Appendix C — state-dict key -> shape of WarpGeneratorS2C/C2D."""
    sh = {"adaptive_matrix_gamma": (512, 512), "adaptive_matrix_beta": (512, 512),
          "flowfield.conv1x1.weight": (2048, 512, 1, 1), "flowfield.conv1x1.bias": (2048,)}
    for k, (ci, co) in enumerate(((512, 256), (256, 128), (128, 64), (64, 32)), start=1):
        p = f"flowfield.resblock{k}."
        sh[p + "conv1.weight"] = (co, ci, 3, 3, 3)
        sh[p + "conv1.bias"] = (co,)
        sh[p + "conv2.weight"] = (co, co, 3, 3, 3)
        sh[p + "conv2.bias"] = (co,)
        for n in ("norm1.", "norm2."):
            sh[p + n + "weight"] = (1, co, 1, 1, 1)
            sh[p + n + "bias"] = (1, co, 1, 1, 1)
            sh[p + n + "group_norm.weight"] = (co,)
            sh[p + n + "group_norm.bias"] = (co,)
        sh[p + "residual_conv.weight"] = (co, ci, 1, 1, 1)
        sh[p + "residual_conv.bias"] = (co,)
    sh["flowfield.conv3x3x3.weight"] = (3, 32, 3, 3, 3)
    sh["flowfield.conv3x3x3.bias"] = (3,)
    sh["flowfield.gn.weight"] = (3,)
    sh["flowfield.gn.bias"] = (3,)
    return sh


def g3d_shapes(in_channels: int = 96) -> Dict[str, tuple]:
    sh = {}
    blocks = (("downsampling.0.", in_channels, 96), ("downsampling.2.", 96, 192),
              ("downsampling.4.", 192, 384), ("downsampling.6.", 384, 768),
              ("upsampling.0.", 768, 384), ("upsampling.2.", 384, 192), ("upsampling.4.", 192, 96))
    for p, ci, co in blocks:
        sh[p + "conv1.weight"] = (co, ci, 3, 3, 3)
        sh[p + "conv1.bias"] = (co,)
        sh[p + "gn1.weight"] = (co,)
        sh[p + "gn1.bias"] = (co,)
        sh[p + "conv2.weight"] = (co, co, 3, 3, 3)
        sh[p + "conv2.bias"] = (co,)
        sh[p + "gn2.weight"] = (co,)
        sh[p + "gn2.bias"] = (co,)
        if ci != co:
            sh[p + "shortcut.weight"] = (co, ci, 1, 1, 1)
            sh[p + "shortcut.bias"] = (co,)
    sh["final_conv.weight"] = (96, 96, 3, 3, 3)
    sh["final_conv.bias"] = (96,)
    return sh


def seeded_state_dict(shapes: Dict[str, tuple], seed: int, prefix: str = "") -> SD:
    """Integer-PRNG weights with PyTorch-default-like magnitudes so activations stay O(1):
    conv/linear weights and biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm scales ~ 1 +- 0.25,
    norm shifts ~ +-0.25; adaptive_matrix ~ U(-1,1)*sqrt(3) (unit variance like randn)."""
    sd = {}
    for i, (k, shp) in enumerate(sorted(shapes.items())):
        s = seed * 1000 + i
        if "adaptive_matrix" in k:
            t = seeded_tensor(shp, s, scale=math.sqrt(3.0))
        elif k.endswith("weight") and len(shp) >= 4 and shp[0] != 1:
            fan_in = int(math.prod(shp[1:]))
            t = seeded_tensor(shp, s, scale=1.0 / math.sqrt(fan_in))
        elif k.endswith("bias") and (k.replace("bias", "weight") in shapes) and len(shapes[k.replace("bias", "weight")]) >= 4 \
                and shapes[k.replace("bias", "weight")][0] != 1:
            fan_in = int(math.prod(shapes[k.replace("bias", "weight")][1:]))
            t = seeded_tensor(shp, s, scale=1.0 / math.sqrt(fan_in))
        elif k.endswith("weight"):
            t = seeded_tensor(shp, s, scale=0.25, shift=1.0)
        else:
            t = seeded_tensor(shp, s, scale=0.25)
        sd[prefix + k] = t
    return sd


def seeded_gbase_hot_state_dict(seed: int) -> SD:
    sd = {}
    sd.update(seeded_state_dict(warp_generator_shapes(), seed + 1, "warp_generator_s2c."))
    sd.update(seeded_state_dict(warp_generator_shapes(), seed + 2, "warp_generator_c2d."))
    sd.update(seeded_state_dict(g3d_shapes(), seed + 3, "G3d."))
    return sd


def seeded_hot_inputs(B: int, seed: int, D: int = 16, H: int = 64, W: int = 64, C: int = 96):
    """SURVEY.md §8(d) shapes with integer-PRNG values: vs~U(-1.7,1.7) (unit variance),
    es,zs,zd likewise, R in (-30,30) deg, t in (-0.17,0.17)."""
    r3 = math.sqrt(3.0)
    return dict(
        vs=seeded_tensor((B, C, D, H, W), seed * 100 + 1, scale=r3),
        es=seeded_tensor((B, 512), seed * 100 + 2, scale=r3),
        zs=seeded_tensor((B, 512), seed * 100 + 3, scale=r3),
        zd=seeded_tensor((B, 512), seed * 100 + 4, scale=r3),
        Rs=seeded_tensor((B, 3), seed * 100 + 5, scale=30.0),
        Rd=seeded_tensor((B, 3), seed * 100 + 6, scale=30.0),
        ts=seeded_tensor((B, 3), seed * 100 + 7, scale=0.17),
        td=seeded_tensor((B, 3), seed * 100 + 8, scale=0.17),
    )
