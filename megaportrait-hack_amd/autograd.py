"""torch.autograd.Function wrappers of the HIP ops (scope row f2: training through the hot path).

The reference trains Gbase with plain PyTorch autograd (`train.py:194-330`: forward, losses, `.backward()`,
optimizer step); these Functions give the same gradients for the G3d building blocks — nn.Conv3d,
nn.GroupNorm (+residual +ReLU), nn.AvgPool3d(2), nn.Upsample(x2 trilinear, align_corners=True)
(model.py:500-528, 571-597) — with every forward and backward pass running in libmphip.so.
No torch eager / CPU fallback: a shape the kernels do not cover raises.
"""
from __future__ import annotations

import torch

from . import ops


def _bwd_pack(conv) -> ops.PackedConv:
    """PackedConv of the flipped/transposed weight (bwd-data as a forward conv), cached on the module like the
    forward pack and rebuilt when the parameter changes."""
    w = conv.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), str(w.device))
    hit = conv.__dict__.get("_mphip_bwd_pack")
    if hit is None or hit[0] != key:
        hit = (key, ops.PackedConv(ops.conv_bwd_data_weight(w), None))
        conv.__dict__["_mphip_bwd_pack"] = hit
    return hit[1]


class Conv3dFn(torch.autograd.Function):
    """y = conv3d(x, W, b, padding=k//2).  `conv` is the nn.Conv3d holding W, b (for the pack caches)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, fwd_pack):
        ctx.conv = conv
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x)
        return ops.conv3d(x, fwd_pack)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        dy = dy.contiguous()
        k = conv.weight.shape[2]
        dx = dw = None
        db, scale = ops.grad_prep(dy, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d_bwd_data(dy, _bwd_pack(conv), scale)
        if ctx.needs_input_grad[1]:
            dw = ops.conv3d_bwd_weight(x, dy, k, scale)
        return dx, dw, db, None, None


class GroupNormFn(torch.autograd.Function):
    """y = act(GroupNorm(x) * gamma + beta (+ residual)), act = ReLU or identity."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, groups, eps, relu):
        x = x.contiguous()
        stats = ops.groupnorm_stats(x, groups, eps)
        y = ops.groupnorm_apply(x, stats, gamma, beta, groups, residual=residual, relu=relu)
        ctx.groups, ctx.relu, ctx.has_res = groups, relu, residual is not None
        ctx.save_for_backward(x, y, stats, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma = ctx.saved_tensors
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        dx, dgamma, dbeta, dres = ops.groupnorm_bwd(x, y, dy.contiguous(), stats, gamma, ctx.groups, ctx.relu, want_res)
        return dx, dgamma, dbeta, dres, None, None, None


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.avgpool2(x.contiguous())

    @staticmethod
    def backward(ctx, dout):
        return ops.avgpool2_bwd(dout.contiguous())


class UpsampleTrilinear2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample_trilinear2(x.contiguous())

    @staticmethod
    def backward(ctx, dout):
        return ops.upsample_trilinear2_bwd(dout.contiguous())


def conv3d(x, conv, fwd_pack):
    return Conv3dFn.apply(x, conv.weight, conv.bias, conv, fwd_pack)


def groupnorm(x, gn, residual=None, relu=False):
    return GroupNormFn.apply(x, gn.weight, gn.bias, residual, gn.num_groups, gn.eps, relu)


def needs_grad(module, *tensors) -> bool:
    return torch.is_grad_enabled() and (any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
                                        or any(p.requires_grad for p in module.parameters()))
