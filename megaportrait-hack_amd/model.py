"""Host-side mirror of the reference's hot-path modules (johndpope/MegaPortrait-hack model.py).

Same class names, constructor signatures, parameter names/shapes (=> identical state-dict keys,
SURVEY.md Appendix C) and forward signatures as the reference, so its callers
(`train.py:194,283`, `inference.py:35`, `PairwiseTransferLoss` model.py:2192-2214) drop in; the
arithmetic runs in libmphip.so (hand-written HIP for gfx950) through the C ABI.  nn.Conv3d /
nn.GroupNorm objects are kept purely as parameter containers (names, shapes, default init,
.to()/.state_dict()); their own forward is never called.

Training: every module here is differentiable — under autograd the same HIP ops run as
torch.autograd Functions (autograd.py) whose backward kernels live in csrc/backward.hip, csrc/conv3d_bwd_f16x3.hip
and csrc/warp.hip (K9/K10); under torch.no_grad() the fused inference path runs.
"""
from __future__ import annotations

import logging
import os

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops

COMPRESS_DIM = 512  # model.py:48
_C2D_EARLY = os.environ.get("MPHIP_C2D_EARLY", "0") == "1"  # dev switch, see GbaseHotSlice._run


def _f32(*tensors):
    """Inputs arrive in fp16/bf16 when an upstream PyTorch module ran under autocast (train.py:188): the HIP path
    computes in fp32, like the reference's fp32 parameters do outside autocast."""
    out = tuple(t.float() if isinstance(t, torch.Tensor) and t.is_floating_point() and t.dtype != torch.float32 else t
                for t in tensors)
    return out[0] if len(out) == 1 else out


def _autocast_policy(fn):
    """The reference runs its generator step under torch.cuda.amp.autocast() (train.py:145,188): conv3d then multiplies f16 operands
    with fp32 accumulation.  Inside such a region the F(2,3) conv launches of the wrapped forward do the same (ops.half_products); outside
    it — inference.py, the graded fp32 configurations — nothing changes.  Inputs are still promoted to fp32 at the boundary (_f32): the
    policy is about the conv arithmetic, activations stay fp32 between kernels."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if not ops.autocast_half():
            return fn(*args, **kwargs)
        with ops.half_products(True):
            return fn(*args, **kwargs)

    return wrapped


class _PackCache:
    """Packed conv weights, rebuilt when the parameter changes (in-place update, load_state_dict or .to()).
    The pack lives ON the conv module (not in a global table keyed by id()), so it dies with the module and
    can never be mistaken for another module's weights after id / address reuse."""

    @staticmethod
    def get(conv: nn.Module) -> ops.PackedConv:
        w, b = conv.weight, conv.bias
        key = ops._pack_cache_key(conv, "_mphip_pack")   # (data_ptr, _version, shape of the weight; the bias; device; epoch)
        hit = conv.__dict__.get("_mphip_pack")
        if not ops.pack_is_current(hit, key):
            hit = (key, ops.PackedConv(w, b))
            conv.__dict__["_mphip_pack"] = hit
        return hit[1]


_packs = _PackCache()


def compute_rt_warp(rotation, translation, invert=False, grid_size=64):
    """model.py:777-809 — rigid warp grid [B,3,G,G,G] (x,y,z channels)."""
    rotation, translation = _f32(rotation, translation)
    zero_em = torch.zeros((rotation.shape[0], 3, 1, 1, 1), dtype=torch.float32, device=rotation.device)
    if torch.is_grad_enabled() and (rotation.requires_grad or translation.requires_grad):
        # un-swapped reference code that calls this under autograd (the reference's own WarpGenerator*.forward, model.py:965,
        # 1016, when only integration.patch_functions ran): rt + resize(0) == rt exactly, with the gradient to R and t
        theta = ag.RtThetaFn.apply(rotation, translation, bool(invert))
        return ag.WarpFieldComposeFn.apply(theta, zero_em, grid_size)
    theta = ops.rt_theta(rotation, translation, invert)
    _, rt, _ = ops.warp_field_compose(theta, zero_em, grid_size, parts=True)
    return rt


def apply_warping_field(v, warp_field):
    """model.py:1028-1065 — same signature and result; one fused HIP kernel (K2); differentiable (autograd.WarpVolumeFn)."""
    v, warp_field = _f32(v, warp_field)
    if torch.is_grad_enabled() and (v.requires_grad or warp_field.requires_grad):
        return ag.WarpVolumeFn.apply(v, warp_field, False)
    return ops.warp_volume(v, warp_field)


def _upsample_trilinear(x, scale_factors):
    """F.interpolate(x, scale_factor=scale_factors, mode='trilinear', align_corners=False) on the HIP kernels."""
    if tuple(float(v) for v in scale_factors) == (1.0, 1.0, 1.0):
        return x
    if torch.is_grad_enabled() and x.requires_grad:
        return ag.UpsampleTrilinearFn.apply(x, tuple(scale_factors))
    return ops.upsample_trilinear(x, scale_factors)


class AdaptiveGroupNorm(nn.Module):
    """model.py:304-316."""

    def __init__(self, num_channels, num_groups=32):
        super().__init__()
        self.num_channels = num_channels
        self.num_groups = num_groups
        self.weight = nn.Parameter(torch.ones(1, num_channels, 1, 1, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_channels, 1, 1, 1))
        self.group_norm = nn.GroupNorm(num_groups, num_channels)

    def forward(self, x):
        x = _f32(x)
        if ag.needs_grad(self, x):
            return ag.adaptive_groupnorm(x, self)
        st = ops.groupnorm_stats(x, self.num_groups, self.group_norm.eps)
        return ops.groupnorm_apply(x, st, self.group_norm.weight, self.group_norm.bias, self.num_groups,
                                   w2=self.weight, b2=self.bias)


class ResBlock3D_Adaptive(nn.Module):
    """model.py:369-408 (the `upsample` flag is never set on the hot path; kept for signature parity)."""

    def __init__(self, in_channels, out_channels, upsample=False, scale_factors=(1, 1, 1)):
        super().__init__()
        self.upsample = upsample
        self.scale_factors = scale_factors
        self.conv1 = nn.Conv3d(in_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv3d(out_channels, out_channels, 3, padding=1)
        self.norm1 = AdaptiveGroupNorm(out_channels)
        self.norm2 = AdaptiveGroupNorm(out_channels)
        if in_channels != out_channels:
            self.residual_conv = nn.Conv3d(in_channels, out_channels, 1)
        else:
            self.residual_conv = nn.Identity()

    def forward(self, x, _up=(1, 1, 1)):
        """`_up`: nearest-upsample factors fused into the block's last elementwise pass (FlowField's nn.Upsample)."""
        out = self._forward(x, _up)
        if self.upsample:   # model.py:404-405 (no module of Gbase sets the flag): trilinear, align_corners=False
            out = _upsample_trilinear(out, self.scale_factors)
        return out

    def _forward(self, x, _up=(1, 1, 1)):
        x = _f32(x)
        if ag.needs_grad(self, x):  # differentiable path: the same ops, unfused, as autograd Functions (autograd.py)
            y = ag.conv3d(x, self.conv1, _packs.get(self.conv1))
            y = ag.adaptive_groupnorm(y, self.norm1, relu=True)
            y = ag.conv3d(y, self.conv2, _packs.get(self.conv2))
            res = x if isinstance(self.residual_conv, nn.Identity) else ag.conv3d(x, self.residual_conv, _packs.get(self.residual_conv))
            y = ag.adaptive_groupnorm(y, self.norm2, residual=res, relu=True)
            return y if tuple(_up) == (1, 1, 1) else ag.UpsampleNearestFn.apply(y, tuple(_up))
        n1, n2 = self.norm1, self.norm2
        res_conv = None if isinstance(self.residual_conv, nn.Identity) else self.residual_conv
        if res_conv is not None and ops.flowfield_conv_gn_ok(tuple(x.shape), self.conv1) and \
                ops.flowfield_conv_gn_ok(tuple(x.shape[:1]) + (self.conv2.weight.shape[1],) + tuple(x.shape[2:]), self.conv2, res_conv):
            # FlowField's four levels: each half of the block is ONE launch (csrc/flowfield.hip)
            a = ops.flowfield_conv_gn(x, self.conv1, n1, relu=True)
            return ops.flowfield_conv_gn(a, self.conv2, n2, res_x=x, res_conv=res_conv, relu=True, up=_up)
        # split-K slabs are summed by the GN kernels below; a direct launch carries the norm's statistics along
        y = ops.conv3d_split(x, _packs.get(self.conv1), gn_groups=n1.num_groups, gn_eps=n1.group_norm.eps)
        tiny = ops.groupnorm_fused_ok(y, n1.num_groups)  # FlowField: statistics + apply in one launch
        if tiny:
            a = ops.groupnorm_small(y, n1.group_norm.weight, n1.group_norm.bias, n1.num_groups, n1.group_norm.eps,
                                    w2=n1.weight, b2=n1.bias, relu=True)
            y = ops.conv3d_split(a, _packs.get(self.conv2))
        else:
            st = ops.groupnorm_stats(y, n1.num_groups, n1.group_norm.eps)
            pc2 = _packs.get(self.conv2)
            if y.splits == 1 and ops.gn_in_conv_ok(y.shape, pc2):  # Eapp's 3D tail: AGN1 + ReLU inside conv2's staging
                y2, st2 = ops.conv3d_gn_in(y.data, st, n1.group_norm.weight, n1.group_norm.bias, n1.num_groups, pc2,
                                           w2=n1.weight, b2=n1.bias, out_gn_groups=n2.num_groups, out_gn_eps=n2.group_norm.eps)
                y = ops.ConvOut(y2, 1, None, y.shape, st2, n2.num_groups)
            else:
                a = ops.groupnorm_apply(y, st, n1.group_norm.weight, n1.group_norm.bias, n1.num_groups, w2=n1.weight,
                                        b2=n1.bias, relu=True)
                y = ops.conv3d_split(a, pc2, gn_groups=n2.num_groups, gn_eps=n2.group_norm.eps)
        res = x if isinstance(self.residual_conv, nn.Identity) else ops.conv3d_split(x, _packs.get(self.residual_conv))
        if tiny:
            return ops.groupnorm_small(y, n2.group_norm.weight, n2.group_norm.bias, n2.num_groups, n2.group_norm.eps,
                                       w2=n2.weight, b2=n2.bias, residual=res, relu=True, up=_up)
        st = ops.groupnorm_stats(y, n2.num_groups, n2.group_norm.eps)
        return ops.groupnorm_apply(y, st, n2.group_norm.weight, n2.group_norm.bias, n2.num_groups, w2=n2.weight,
                                   b2=n2.bias, residual=res, relu=True, up=_up)


class FlowField(nn.Module):
    """model.py:415-471: [B,512,1,1] -> [B,3,16,16,16] in [0,1)."""

    _UPS = ((2, 2, 2), (2, 2, 2), (1, 2, 2), (1, 2, 2))

    def __init__(self):
        super().__init__()
        self.conv1x1 = nn.Conv2d(512, 2048, kernel_size=1)
        self.resblock1 = ResBlock3D_Adaptive(in_channels=512, out_channels=256)
        self.upsample1 = nn.Upsample(scale_factor=(2, 2, 2))
        self.resblock2 = ResBlock3D_Adaptive(in_channels=256, out_channels=128)
        self.upsample2 = nn.Upsample(scale_factor=(2, 2, 2))
        self.resblock3 = ResBlock3D_Adaptive(in_channels=128, out_channels=64)
        self.upsample3 = nn.Upsample(scale_factor=(1, 2, 2))
        self.resblock4 = ResBlock3D_Adaptive(in_channels=64, out_channels=32)
        self.upsample4 = nn.Upsample(scale_factor=(1, 2, 2))
        self.conv3x3x3 = nn.Conv3d(32, 3, kernel_size=3, padding=1)
        self.gn = nn.GroupNorm(1, 3)
        self.tanh = nn.Tanh()

    def _conv1x1_kn(self):
        """conv1x1.weight [2048,512,1,1] re-laid as [K=512][N=2048] (N contiguous) for the coalesced matmul kernel;
        rebuilt when the parameter changes."""
        w = self.conv1x1.weight
        key = (w.data_ptr(), w._version, str(w.device), ops.weight_epoch())
        hit = self.__dict__.get("_w_kn")
        if hit is None or hit[0] != key or ops.repacking():
            hit = (key, w.detach().reshape(2048, 512).t().contiguous())
            self.__dict__["_w_kn"] = hit
        return hit[1]

    def _head_kn(self, gamma):
        """(z+e) @ Gamma followed by the 1x1 conv on the 1x1 map (model.py:945-957 -> 446) is one linear map of (z+e):
        Gamma @ W^T, [512][2048], built once per weight version — one dense product per step instead of two dependent ones
        at the head of the latency-bound generator chain."""
        w = self.conv1x1.weight
        key = (w.data_ptr(), w._version, gamma.data_ptr(), gamma._version, str(w.device), ops.weight_epoch())
        hit = self.__dict__.get("_w_head")
        if hit is None or hit[0] != key or ops.repacking():
            hit = (key, ops.small_gemm(gamma.detach(), self._conv1x1_kn()))
            self.__dict__["_w_head"] = hit
        return hit[1]

    def forward_from_codes(self, z, e, gamma):
        """Inference entry used by the warp generators: FlowField((z+e) @ Gamma) with the two dense products merged."""
        z, e = _f32(z, e)
        b = z.shape[0]
        x = ops.add_matmul(z.reshape(b, 512), e.reshape(b, 512), self._head_kn(gamma), self.conv1x1.bias)
        return self._tail(x.view(b, 512, 4, 1, 1), False)

    def forward(self, zs, adaptive_gamma=0, adaptive_beta=0):  # last two ignored, as in the reference
        zs = _f32(zs)
        train = ag.needs_grad(self, zs)
        b = zs.shape[0]
        s = zs.reshape(b, 512)
        if train:
            x = ag.Conv1x1OnVectorFn.apply(s, self.conv1x1.weight, self.conv1x1.bias, self._conv1x1_kn())
        else:
            x = ops.add_matmul(s, None, self._conv1x1_kn(), self.conv1x1.bias)  # 1x1 conv on a 1x1 map == s @ W^T + b
        return self._tail(x.view(b, 512, 4, 1, 1), train)  # model.py:425: channel c*4+d -> (c,d)

    def _tail(self, x, train):
        for blk, up in zip((self.resblock1, self.resblock2, self.resblock3, self.resblock4), self._UPS):
            x = blk(x, _up=up)  # nn.Upsample (nearest, model.py:450-457) fused into the block's last pass
        if train:
            x = ag.conv3d(x, self.conv3x3x3, _packs.get(self.conv3x3x3))
            x = ag.groupnorm(x, self.gn, relu=True, tanh=True)
            assert x.shape[1] == 3, f"Expected 3 channels after conv3x3x3, got {x.shape[1]}"
            return x
        if ops.flowfield_out_ok(tuple(x.shape), self.conv3x3x3) and self.gn.num_groups == 1:
            return ops.flowfield_out(x, self.conv3x3x3, self.gn)   # direct 3-channel conv + one normalising pass (csrc/flowfield.hip)
        x = ops.conv3d_split(x, _packs.get(self.conv3x3x3))
        if ops.groupnorm_fused_ok(x, 1):
            x = ops.groupnorm_small(x, self.gn.weight, self.gn.bias, 1, self.gn.eps, relu=True, tanh=True)
        else:
            st = ops.groupnorm_stats(x, 1, self.gn.eps)
            x = ops.groupnorm_apply(x, st, self.gn.weight, self.gn.bias, 1, relu=True, tanh=True)
        assert x.shape[1] == 3, f"Expected 3 channels after conv3x3x3, got {x.shape[1]}"
        return x


class _WarpGenerator(nn.Module):
    _INVERT = False

    def __init__(self, num_channels):
        super().__init__()
        self.flowfield = FlowField()
        self.num_channels = COMPRESS_DIM
        # Registered parameters on every device (the reference's `nn.Parameter(...).to(device)` only
        # registers them on CPU hosts, model.py:934-935; load_gbase_state_dict() accepts both layouts).
        self.adaptive_matrix_gamma = nn.Parameter(torch.randn(self.num_channels, self.num_channels))
        self.adaptive_matrix_beta = nn.Parameter(torch.randn(self.num_channels, self.num_channels))

    def forward(self, R, t, z, e):
        assert R.shape == (z.shape[0], 3), f"Expected R shape (batch_size, 3), got {R.shape}"
        assert t.shape == (z.shape[0], 3), f"Expected t shape (batch_size, 3), got {t.shape}"
        assert z.shape == e.shape, f"Expected z and e to have the same shape, got {z.shape} and {e.shape}"
        R, t, z, e = _f32(R, t, z, e)
        if ag.needs_grad(self, R, t, z, e):
            s = ag.AddMatmulFn.apply(z, e, self.adaptive_matrix_gamma)
            em = self.flowfield(s.unsqueeze(-1).unsqueeze(-1), 0, 0)
            theta = ag.RtThetaFn.apply(R, t, self._INVERT)
            return ag.WarpFieldComposeFn.apply(theta, em, 64)
        em = self.flowfield.forward_from_codes(z, e, self.adaptive_matrix_gamma)  # FlowField((z+e) @ Gamma), model.py:945-957
        theta = ops.rt_theta(R, t, self._INVERT)
        return ops.warp_field_compose(theta, em, 64)


class WarpGeneratorS2C(_WarpGenerator):
    """model.py:927-975 (rigid part inverted)."""
    _INVERT = True


class WarpGeneratorC2D(_WarpGenerator):
    """model.py:978-1024."""
    _INVERT = False


class ResBlock3D(nn.Module):
    """model.py:500-528."""

    def __init__(self, in_channels, out_channels, upsample=False, scale_factors=(1, 1, 1)):
        super().__init__()
        self.upsample = upsample
        self.scale_factors = scale_factors
        self.conv1 = nn.Conv3d(in_channels, out_channels, kernel_size=3, padding=1)
        self.gn1 = nn.GroupNorm(num_groups=32, num_channels=out_channels)
        self.conv2 = nn.Conv3d(out_channels, out_channels, kernel_size=3, padding=1)
        self.gn2 = nn.GroupNorm(num_groups=32, num_channels=out_channels)
        self.shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else nn.Identity()

    def _forward_train(self, x):
        """Differentiable path (scope row f2): the same ops, unfused, as torch.autograd Functions whose forward and
        backward both run in libmphip.so (autograd.py)."""
        identity = x if isinstance(self.shortcut, nn.Identity) else ag.conv3d(x, self.shortcut, _packs.get(self.shortcut))
        y = ag.conv3d(x, self.conv1, _packs.get(self.conv1))
        y = ag.groupnorm(y, self.gn1, relu=True)
        y = ag.conv3d(y, self.conv2, _packs.get(self.conv2))
        return ag.groupnorm(y, self.gn2, residual=identity, relu=True)

    def forward(self, x, _pool_after: bool = False, _after_conv1=None):
        """`_after_conv1`: host-side hook called right after conv1 has been launched (GbaseHotSlice issues the side-stream
        generator there, once the GPU has a long kernel queued)."""
        out = self._forward(x, _pool_after, _after_conv1)
        if self.upsample:   # model.py:525-526 (G3d never sets the flag): trilinear, align_corners=False
            out = _upsample_trilinear(out, self.scale_factors)
        return out

    def _forward(self, x, _pool_after: bool = False, _after_conv1=None):
        x = _f32(x)
        if ag.needs_grad(self, x):
            y = self._forward_train(x)
            if _after_conv1 is not None:
                _after_conv1()
            return ag.AvgPool2Fn.apply(y) if _pool_after else y
        identity = x if isinstance(self.shortcut, nn.Identity) else ops.conv3d_split(x, _packs.get(self.shortcut))
        y = ops.conv3d_split(x, _packs.get(self.conv1), gn_groups=32, gn_eps=self.gn1.eps)  # + GN1's statistics
        if _after_conv1 is not None:
            _after_conv1()
        st = ops.groupnorm_stats(y, 32, self.gn1.eps)
        pc2 = _packs.get(self.conv2)
        if y.splits == 1 and ops.gn_in_conv_ok(y.shape, pc2):
            # GN1 + ReLU folded into conv2's input staging: the normalised tensor never touches HBM
            y2, st2 = ops.conv3d_gn_in(y.data, st, self.gn1.weight, self.gn1.bias, 32, pc2, out_gn_groups=32,
                                       out_gn_eps=self.gn2.eps)
            y = ops.ConvOut(y2, 1, None, y.shape, st2, 32)
        else:
            a = ops.groupnorm_apply(y, st, self.gn1.weight, self.gn1.bias, 32, relu=True)
            y = ops.conv3d_split(a, pc2, gn_groups=32, gn_eps=self.gn2.eps)
        st = ops.groupnorm_stats(y, 32, self.gn2.eps)
        return ops.groupnorm_apply(y, st, self.gn2.weight, self.gn2.bias, 32, residual=identity, relu=True,
                                   pool2=_pool_after)


class G3d(nn.Module):
    """model.py:571-597.  nn.Sequential indices give the reference's state-dict names."""

    def __init__(self, in_channels):
        super().__init__()
        self.downsampling = nn.Sequential(
            ResBlock3D(in_channels, 96), nn.AvgPool3d(kernel_size=2, stride=2),
            ResBlock3D(96, 192), nn.AvgPool3d(kernel_size=2, stride=2),
            ResBlock3D(192, 384), nn.AvgPool3d(kernel_size=2, stride=2),
            ResBlock3D(384, 768),
        )
        self.upsampling = nn.Sequential(
            ResBlock3D(768, 384), nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True),
            ResBlock3D(384, 192), nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True),
            ResBlock3D(192, 96), nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True),
        )
        self.final_conv = nn.Conv3d(96, 96, kernel_size=3, padding=1)

    @_autocast_policy
    def forward(self, x, _after_first_conv=None, _final_roi=None):
        """`_final_roi`: sample boxes of the warp that is the ONLY reader of the result (GbaseHotSlice under autograd): final_conv
        is evaluated on the tiles they touch, the rest of the returned tensor is uninitialised (ops.conv3d_roi)."""
        x = _f32(x)
        train = ag.needs_grad(self, x)
        up = ag.UpsampleTrilinear2Fn.apply if train else ops.upsample_trilinear2
        d = self.downsampling
        x = d[0](x, _pool_after=True, _after_conv1=_after_first_conv)   # AvgPool3d fused into the block's last elementwise pass (inference)
        x = d[2](x, _pool_after=True)
        x = d[4](x, _pool_after=True)
        x = d[6](x)
        u = self.upsampling
        x = up(u[0](x))
        x = up(u[2](x))
        x = up(u[4](x))
        if train:
            return ag.conv3d(x, self.final_conv, _packs.get(self.final_conv), roi=_final_roi)
        return ops.conv3d(x, _packs.get(self.final_conv))


class Eapp3DTail(nn.Module):
    """Scope row f1: the 3D tail of Eapp, model.py:217-226 + 271-290.  Attribute names are
    the reference Eapp's, so `appearanceEncoder.resblock3D_*` checkpoint keys load into this module directly.
    The reference assigns `resblock3D_96_2` twice (model.py:218,225): five blocks exist, one is applied twice."""

    _ORDER = ("resblock3D_96", "resblock3D_96_2", "resblock3D_96_1", "resblock3D_96_1_2", "resblock3D_96_2",
              "resblock3D_96_2_2")
    max_frames_per_pass = 64

    def __init__(self):
        super().__init__()
        self.resblock3D_96 = ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_1 = ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_1_2 = ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_2 = ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_2_2 = ResBlock3D_Adaptive(in_channels=96, out_channels=96)

    @_autocast_policy
    def forward(self, out):
        """out: Eapp's conv_1 output [B,1536,H,W] (model.py:268) or the reshaped volume [B,96,16,H,W]."""
        out = _f32(out)
        vs = out.view(out.size(0), 96, 16, *out.shape[2:]) if out.dim() == 4 else out  # model.py:271
        if vs.shape[0] > self.max_frames_per_pass and not ag.needs_grad(self, vs):
            # (the conv kernels' 2 GiB buffer resource: see _HotSliceRunner.max_frames_per_pass; frames are independent)
            step = int(self.max_frames_per_pass)
            return torch.cat([self.forward(vs[i:i + step]) for i in range(0, vs.shape[0], step)], dim=0)
        for name in self._ORDER:
            vs = getattr(self, name)(vs)
        return vs


class G2dHead(nn.Module):
    """Scope row f3: the entry of G2d, model.py:718-719 + 756-757 — `reshape` Conv2d(96,1536,1)
    followed directly by `conv1x1` Conv2d(1536,512,1).  There is no nonlinearity between them, so at inference the two
    collapse into ONE 96->512 product (W = W2 @ W1, b = W2 @ b1 + b2): 19x fewer FLOPs and the 25 MB/frame 1536-channel
    intermediate never exists.  Attribute names are G2d's, so `G2d.reshape.*` / `G2d.conv1x1.*` checkpoint keys load
    here.  Under autograd the two convs run separately (their parameters get their own gradients)."""

    def __init__(self):
        super().__init__()
        self.reshape = nn.Conv2d(96, 1536, kernel_size=1)
        self.conv1x1 = nn.Conv2d(1536, 512, kernel_size=1)

    def _fused_pack(self) -> ops.PackedConv:
        ps = (self.reshape.weight, self.reshape.bias, self.conv1x1.weight, self.conv1x1.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps) + (str(ps[0].device), ops.weight_epoch())
        hit = self.__dict__.get("_fused")
        if hit is None or hit[0] != key or ops.repacking():
            w1, b1, w2, b2 = (p.detach() for p in ps)
            w = ops.small_gemm(w2.reshape(512, 1536), w1.reshape(1536, 96))                       # [512, 96]
            b = ops.small_gemm(b1.reshape(1, 1536), w2.reshape(512, 1536), trans_b=True, bias=b2)  # [1, 512]
            hit = (key, ops.PackedConv(w.view(512, 96, 1, 1, 1), b.view(512)))
            self.__dict__["_fused"] = hit
        return hit[1]

    def forward(self, x):
        """x: the hot slice's output [B,96,H,W] (model.py:1171) -> [B,512,H,W], the input of G2d's ResBlock2D stack."""
        x = _f32(x)
        b, c, h, w = x.shape
        x5 = x.reshape(b, c, 1, h, w)
        if ag.needs_grad(self, x):
            y = ag.conv3d(x5, self.reshape, _packs.get(self.reshape))
            y = ag.conv3d(y, self.conv1x1, _packs.get(self.conv1x1))
        else:
            y = ops.conv3d(x5, self._fused_pack())
        return y.reshape(b, 512, h, w)


class _HotSliceRunner:
    """model.py:1151-1171 over `self.warp_generator_s2c`, `self.warp_generator_c2d`, `self.G3d` — shared by
    GbaseHotSlice (the slice alone) and gbase.Gbase (the orchestrator)."""

    # The C2D generator depends only on (Rd, td, zd, es): its ~25 small, latency-bound launches run on
    # a second HIP stream underneath G3d's MFMA-bound kernels instead of in front of the last warp.
    overlap_generators = True
    # Frames per pass of the slice: the conv kernels address their input through one 2 GiB buffer resource (a [B,96,16,64,64]
    # volume is 25 MB per frame: 85 frames); larger batches run as consecutive passes of this many frames (frames are
    # independent: same results).  Under autograd the batch is not split (a training batch of that size does not fit anyway).
    max_frames_per_pass = 64

    def _side_stream(self, main):
        """One helper stream per caller stream (callers may keep several batches in flight on their own streams)."""
        table = self.__dict__.setdefault("_side_streams", {})
        key = (main.device, main.cuda_stream)
        st = table.get(key)
        if st is None:
            st = table[key] = torch.cuda.Stream(device=main.device)
        return st

    # Inference goes through the C-side plan (csrc/plan.hip, include/mphip.h "one-call entries"): ONE ctypes call per step
    # issues the same launches in the same order as `_run_python` below (bitwise identical results, tests/test_gpu_plan.py);
    # a single frame is launch-bound and the ~135 Python/ctypes round trips cost more than the kernels.  MPHIP_C_PLAN=0 or
    # `use_c_plan = False` selects the per-op schedule (also taken under autograd and when a measurement hook is installed).
    use_c_plan = os.environ.get("MPHIP_C_PLAN", "1") != "0"
    # G3d's last upsample + conv only where the final warp reads them (MPHIP_FULL_FINAL_CONV=1: everywhere; same output bits)
    full_final_conv = os.environ.get("MPHIP_FULL_FINAL_CONV", "0") == "1"

    def _plan_for(self, vs):
        from . import plan as _plan

        # one plan per caller stream: a plan owns ONE side stream and one fork/join event pair, so callers that keep several batches in
        # flight on several streams get independent generator lanes (and their own packed weights: ~230 MB per plan)
        shape_key = (tuple(vs.shape[1:]), vs.device, bool(self.overlap_generators), bool(self.full_final_conv))
        key = shape_key + (torch.cuda.current_stream(vs.device).cuda_stream,)
        table = self.__dict__.setdefault("_plans", {})
        if torch.cuda.is_current_stream_capturing():
            # torch.cuda.graph() switches to its own capture stream: a plan made HERE would hipMalloc, copy tables synchronously and pack
            # weights inside the capture (illegal, or recorded into every replay).  Take the plan that was warmed up for this shape —
            # its launches go to whatever stream is current — and refuse to build one while capturing.  
            for k in reversed(list(table)):
                if k[:len(shape_key)] == shape_key:
                    return table[k]
            raise RuntimeError("GbaseHotSlice: no warmed-up plan for this shape while a stream is capturing — run one forward of the same "
                               "shape before torch.cuda.graph() (model.GraphedHotSlice does)")
        pl = table.pop(key, None)
        if pl is None:
            pl = _plan.HotSlicePlan(self, dims=tuple(vs.shape[1:]), single_stream=not self.overlap_generators,
                                    full_final_conv=self.full_final_conv)
        table[key] = pl   # (most recently used last)
        while len(table) > self._MAX_PLANS:   # a caller that keeps making new streams must not keep every plan's packed weights
            torch.cuda.synchronize(vs.device)
            table.pop(next(iter(table))).close()
        return pl

    _MAX_PLANS = 6

    @_autocast_policy
    def _run(self, vs, es, Rs, ts, zs, Rd, td, zd, check_shape: bool):
        vs, es, Rs, ts, zs, Rd, td, zd = _f32(vs, es, Rs, ts, zs, Rd, td, zd)
        if (self.use_c_plan and vs.shape[0] > 0 and vs.dim() == 5 and vs.shape[1] == 96 and ops._conv_hook is None
                and ops._GNIN_ENABLED and ops._GN_SMALL_ENABLED and ops._RANGES_ENABLED   # (dev A/B switches act on the per-op path)
                and not ag.needs_grad(self, vs, es, Rs, ts, zs, Rd, td, zd)
                and all(v % 8 == 0 for v in vs.shape[2:])):
            if check_shape:
                assert vs.shape[1:] == (96, 16, 64, 64), f"Expected vc shape (_, 96, 16, 64, 64), got {vs.shape}"
            return self._plan_for(vs).forward(vs, es, Rs, ts, zs, Rd, td, zd)
        return self._run_python(vs, es, Rs, ts, zs, Rd, td, zd, check_shape)

    def _run_python(self, vs, es, Rs, ts, zs, Rd, td, zd, check_shape: bool):
        vs, es, Rs, ts, zs, Rd, td, zd = _f32(vs, es, Rs, ts, zs, Rd, td, zd)
        if vs.shape[0] == 0:  # an empty frame shard (dp.shard_inputs with more ranks than frames): nothing to launch
            out = vs.new_zeros((0, vs.shape[1]) + tuple(vs.shape[3:]))
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                # training: stay connected to the parameters (zero gradients), so this rank's loss.backward() runs and it
                # enters the gradient all-reduce with the others instead of leaving them blocked in the collective
                # (multiply-free: `p.sum() * 0` would turn one Inf/NaN weight into a NaN loss on this rank and, through the
                #  all-reduce, on every rank; an empty slice sums to an exact 0 with zero gradients and launches no reduction)
                out = out + sum(p.reshape(-1)[:0].sum() for p in self.parameters() if p.requires_grad)
            return out
        train = ag.needs_grad(self, vs, es, Rs, ts, zs, Rd, td, zd)
        if not train and vs.shape[0] > self.max_frames_per_pass:
            step = int(self.max_frames_per_pass)
            return torch.cat([self._run_python(*(t[i:i + step] for t in (vs, es, Rs, ts, zs, Rd, td, zd)), check_shape)
                              for i in range(0, vs.shape[0], step)], dim=0)
        main = torch.cuda.current_stream(vs.device)
        # training: one stream (autograd replays each op's backward on its forward stream; the overlap is an inference trick)
        side = self._side_stream(main) if (self.overlap_generators and not train) else None
        if side is not None:
            side.wait_stream(main)  # inputs produced on the main stream are visible
        # Critical path first, in HOST order too: S2C, warp #1 and G3d's first (0.6 ms) conv are launched before the host
        # spends its ~0.3 ms issuing the C2D generator's ~25 launches on the side stream.  Issued earlier, they leave the
        # main stream idle between S2C and warp #1 whenever the host is not running ahead (first step, profiled runs);
        # in the steady-state throughput loop the host is ahead anyway and the order is worth +0.3 % only.
        w_s2c = self.warp_generator_s2c(Rs, ts, zs, es)
        c2d = {}

        def issue_c2d():
            with torch.cuda.stream(side):
                c2d["w"] = self.warp_generator_c2d(Rd, td, zd, es)

        early = side is not None and _C2D_EARLY  # dev switch: the old issue order, for same-box A/B runs
        if early:
            issue_c2d()
        vc = apply_warping_field(vs, w_s2c)
        if check_shape:
            assert vc.shape[1:] == (96, 16, 64, 64), f"Expected vc shape (_, 96, 16, 64, 64), got {vc.shape}"
        roi = None
        if train and not self.full_final_conv and isinstance(self.G3d, G3d):
            # training: the final warp reads (and sends gradient to) only the voxels inside its sample boxes — final_conv's
            # forward is evaluated there (the C-side plan does the same at inference); backward kernels unchanged
            c2d["w"] = self.warp_generator_c2d(Rd, td, zd, es)
            with torch.no_grad():
                roi = ops.warp_sample_box(ops.warp_coords(c2d["w"].detach(), *vc.shape[2:]))
        vc2d = self.G3d(vc, _after_first_conv=issue_c2d if (side is not None and not early) else None, **({"_final_roi": roi} if roi is not None else {}))
        if side is not None:
            w_c2d = c2d["w"]
            main.wait_stream(side)
            w_c2d.record_stream(main)
        elif "w" in c2d:
            w_c2d = c2d["w"]
        else:
            w_c2d = self.warp_generator_c2d(Rd, td, zd, es)
        # apply_warping_field + torch.sum(dim=2) (model.py:1167-1171) in one kernel (K3)
        if train:
            return ag.WarpVolumeFn.apply(vc2d, w_c2d, True)
        return ops.warp_volume_dsum(vc2d, w_c2d)



class GbaseHotSlice(_HotSliceRunner, nn.Module):
    """The slice of Gbase.forward between the 2D encoders and G2d (model.py:1151-1171), with the
    reference's attribute names so a Gbase checkpoint's `warp_generator_s2c.*`,
    `warp_generator_c2d.*` and `G3d.*` keys load unchanged."""

    def __init__(self):
        super().__init__()
        self.warp_generator_s2c = WarpGeneratorS2C(num_channels=512)
        self.warp_generator_c2d = WarpGeneratorC2D(num_channels=512)
        self.G3d = G3d(in_channels=96)

    def forward(self, vs, es, Rs, ts, zs, Rd, td, zd):
        return self._run(vs, es, Rs, ts, zs, Rd, td, zd, True)

    def forward_any_size(self, vs, es, Rs, ts, zs, Rd, td, zd):
        """Same graph without the 512^2-only shape assert (small parity cases)."""
        return self._run(vs, es, Rs, ts, zs, Rd, td, zd, False)


class GraphedHotSlice:
    """Replays the hot slice as one hipGraph: the ~130 stream-ordered launches of a step (two streams) are
    captured once for fixed input shapes and re-launched with a single hipGraphLaunch, which removes the
    per-launch host cost (Python + ctypes + hipLaunchKernel ~ 15 us each) from the critical path.
    Inputs are copied into static buffers; the output tensor is reused between calls."""

    def __init__(self, hot: "GbaseHotSlice", example_inputs: dict, any_size: bool = False, warmup: int = 2):
        self.hot = hot
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        fn = hot.forward_any_size if any_size else hot.forward
        with torch.no_grad():
            side = torch.cuda.Stream(device=next(iter(example_inputs.values())).device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # warm up off the default stream (packs weights, fills caches)
                for _ in range(warmup):
                    fn(**self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            ops.begin_capture()   # range descriptors measured on the warm-up batch are not frozen into the graph
            with torch.cuda.graph(self.graph):
                self.static_out = fn(**self.static_in)
            self.memset_nodes_replaced = ops.finish_graph_capture(self.graph)   # (memset nodes -> kernel nodes, then instantiate)

    def __call__(self, **inputs):
        for k, v in inputs.items():
            if v.data_ptr() != self.static_in[k].data_ptr():
                self.static_in[k].copy_(v)
        self.graph.replay()
        return self.static_out


def load_hot_state_dict(module: nn.Module, state_dict, strict: bool = True):
    """Loads a (sub)set of a reference Gbase state-dict.  Reference checkpoints built on a GPU host
    lack `adaptive_matrix_gamma/beta` (model.py:934-935 quirk): those keys are then left at their
    current values and reported."""
    own = module.state_dict()
    filtered = {k: v for k, v in state_dict.items() if k in own}
    missing = [k for k in own if k not in filtered]
    tolerated = [k for k in missing if "adaptive_matrix_" in k]
    hard_missing = [k for k in missing if k not in tolerated]
    if strict and hard_missing:
        raise KeyError(f"missing keys: {hard_missing[:8]}{'...' if len(hard_missing) > 8 else ''}")
    module.load_state_dict(filtered, strict=False)
    if tolerated:
        logging.warning("checkpoint has no %s (GPU-built reference model); kept current values", tolerated)
    return hard_missing, tolerated
