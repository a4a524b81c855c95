"""torch.autograd.Function wrappers of the HIP ops (scope row f2: training through the hot path).

The reference trains Gbase with plain PyTorch autograd (`train.py:194-330`: forward, losses, `.backward()`,
optimizer step); these Functions give the same gradients for the G3d building blocks — nn.Conv3d,
nn.GroupNorm (+residual +ReLU), nn.AvgPool3d(2), nn.Upsample(x2 trilinear, align_corners=True)
(model.py:500-528, 571-597) — with every forward and backward pass running in libmphip.so.
No torch eager / CPU fallback: a shape the kernels do not cover raises.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import ops

# train.py:188 runs the generator under torch.cuda.amp.autocast(): the HIP path computes in fp32, so
# every Function casts floating-point inputs back to fp32 and runs with autocast disabled
_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


def _bwd_pack(conv) -> ops.PackedConv:
    """PackedConv of the flipped/transposed weight (bwd-data as a forward conv), cached on the module like the
    forward pack and rebuilt when the parameter changes."""
    w = conv.weight
    key = ops._pack_cache_key(conv, "_mphip_bwd_pack")
    hit = conv.__dict__.get("_mphip_bwd_pack")
    if not ops.pack_is_current(hit, key):
        fwd = conv.__dict__.get("_mphip_pack")  # the forward pack — usable only if it was made from these very weights
        same = fwd is not None and fwd[0][0] == w.data_ptr() and fwd[0][1] == w._version and fwd[0][-1] == ops.weight_epoch()
        hit = (key, ops.PackedConv(w, None, transposed=True, header_from=fwd[1] if same else None))
        conv.__dict__["_mphip_bwd_pack"] = hit
    return hit[1]


class Conv3dFn(torch.autograd.Function):
    """y = conv3d(x, W, b, padding=k//2).  `conv` is the nn.Conv3d holding W, b (for the pack caches)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, weight, bias, conv, fwd_pack, roi=None):
        ctx.conv = conv
        ctx.has_bias = bias is not None
        # weight is saved too (not only read from the module in backward): autograd's version check then catches an
        # in-place weight update between forward and backward
        ctx.save_for_backward(x, weight)
        # roi: the boxes of voxels the ONLY consumer (the final warp) reads — the forward is evaluated there and nowhere else
        # (ops.conv3d_roi); the backward is restricted the same way (see backward)
        y = ops.conv3d(x, fwd_pack) if roi is None else ops.conv3d_roi(x, fwd_pack, roi)
        ctx.roi = roi
        ctx.x_range = ops.tensor_range(x)   # the f16x3 operand scale the forward used for x: bwd-weight reuses it
        ctx.half = ops.half_products_active()   # the forward ran under the autocast policy (model._autocast_policy set the library's flag; torch's
                                                # own autocast state is off inside custom_fwd): its backward convs do too, on autograd's thread
        return y

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dy):
        x, _weight = ctx.saved_tensors
        conv = ctx.conv
        dy = dy.contiguous()
        k = conv.weight.shape[2]
        dx = dw = None
        db, scale = ops.grad_prep(dy, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        # ctx.roi: the forward was demand-driven for a gather — that gather's gradient dy is exactly zero outside the same boxes
        # (mphip_warp_volume_bwd zero-fills, then writes the cells the samples touch), so both products are restricted to them
        if ctx.needs_input_grad[0]:
            with ops.half_products(ctx.half):
                dx = ops.conv3d_bwd_data(dy, _bwd_pack(conv), scale, roi=ctx.roi)
        if ctx.needs_input_grad[1]:
            with ops.half_products(ctx.half):   # (the 3x3x3 f16x3 bwd-weight kernel follows the policy too: one product per multiply)
                dw = ops.conv3d_bwd_weight(x, dy, k, scale, x_range=ctx.x_range, roi=ctx.roi).view(conv.weight.shape)  # (a 1x1 Conv2d's weight is 4-D)
        return dx, dw, db, None, None, None


class GroupNormFn(torch.autograd.Function):
    """y = act(GroupNorm(x) * gamma + beta [then * w2 + b2] (+ residual)); act: 0 none, 1 ReLU, 2 tanh(ReLU(.)).
    w2/b2 = AdaptiveGroupNorm's second affine ([1,C,1,1,1], model.py:304-316) or None."""

    @staticmethod
    @_fwd
    def forward(ctx, x, gamma, beta, w2, b2, residual, groups, eps, act):
        x = x.contiguous()
        stats = ops.groupnorm_stats(x, groups, eps)
        y = ops.groupnorm_apply(x, stats, gamma, beta, groups, w2=w2, b2=b2, residual=residual, relu=act >= 1, tanh=act == 2)
        ctx.groups, ctx.act, ctx.has_res, ctx.has_w2 = groups, act, residual is not None, w2 is not None
        ctx.w2_shape = None if w2 is None else tuple(w2.shape)
        if w2 is None:
            ctx.save_for_backward(x, y, stats, gamma)
        else:
            ctx.save_for_backward(x, y, stats, gamma, beta, w2)
        return y

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dy):
        want_res = ctx.has_res and ctx.needs_input_grad[5]
        if ctx.has_w2:
            x, y, stats, gamma, beta, w2 = ctx.saved_tensors
            dx, dgamma, dbeta, dres, dw2, db2 = ops.groupnorm_bwd(x, y, dy.contiguous(), stats, gamma, ctx.groups, ctx.act,
                                                                  want_res, beta=beta, w2=w2)
            return dx, dgamma, dbeta, dw2.view(ctx.w2_shape), db2.view(ctx.w2_shape), dres, None, None, None
        x, y, stats, gamma = ctx.saved_tensors
        dx, dgamma, dbeta, dres = ops.groupnorm_bwd(x, y, dy.contiguous(), stats, gamma, ctx.groups, ctx.act, want_res)
        return dx, dgamma, dbeta, None, None, dres, None, None, None


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, x):
        return ops.avgpool2(x.contiguous())

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dout):
        return ops.avgpool2_bwd(dout.contiguous())


class UpsampleTrilinear2Fn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, x):
        return ops.upsample_trilinear2(x.contiguous())

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dout):
        return ops.upsample_trilinear2_bwd(dout.contiguous())


class UpsampleTrilinearFn(torch.autograd.Function):
    """F.interpolate(scale_factor=(sD,sH,sW), 'trilinear', align_corners=False): the residual blocks' `upsample=True` branch."""

    @staticmethod
    @_fwd
    def forward(ctx, x, scale):
        ctx.scale = tuple(scale)
        return ops.upsample_trilinear(x.contiguous(), ctx.scale)

    @staticmethod
    @once_differentiable
    @_bwd
    def backward(ctx, dout):
        return ops.upsample_trilinear_bwd(dout.contiguous(), ctx.scale), None


def conv3d(x, conv, fwd_pack, roi=None):
    return Conv3dFn.apply(x, conv.weight, conv.bias, conv, fwd_pack, roi)


def groupnorm(x, gn, residual=None, relu=False, tanh=False):
    return GroupNormFn.apply(x, gn.weight, gn.bias, None, None, residual, gn.num_groups, gn.eps, 2 if tanh else int(relu))


def adaptive_groupnorm(x, agn, residual=None, relu=False):
    """AdaptiveGroupNorm (model.py:304-316) (+ residual + ReLU of ResBlock3D_Adaptive, model.py:390-403)."""
    gn = agn.group_norm
    return GroupNormFn.apply(x, gn.weight, gn.bias, agn.weight, agn.bias, residual, gn.num_groups, gn.eps, int(relu))


def needs_grad(module, *tensors) -> bool:
    return torch.is_grad_enabled() and (any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
                                        or any(p.requires_grad for p in module.parameters()))


class WarpVolumeFn(torch.autograd.Function):
    """apply_warping_field (model.py:1028-1065); dsum=True fuses the torch.sum(dim=2) of model.py:1171."""

    @staticmethod
    @_fwd
    def forward(ctx, v, field, dsum):
        ctx.dsum = dsum
        ctx.save_for_backward(v, field)
        return ops.warp_volume_dsum(v, field) if dsum else ops.warp_volume(v, field)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dout):
        v, field = ctx.saved_tensors
        dv, dfield = ops.warp_volume_bwd(v, field, dout.contiguous(), ctx.dsum, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dv, dfield, None


class WarpFieldComposeFn(torch.autograd.Function):
    """rt + em64 of the warp generators: F.affine_grid(theta) + trilinear(em -> G^3) (model.py:804-806, 971-973)."""

    @staticmethod
    @_fwd
    def forward(ctx, theta, em, grid_size):
        ctx.em_shape = tuple(em.shape)
        return ops.warp_field_compose(theta, em, grid_size)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dw):
        dtheta, dem = ops.warp_field_compose_bwd(dw.contiguous(), ctx.em_shape, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dtheta, dem, None


class RtThetaFn(torch.autograd.Function):
    """theta = rows 0..2 of [R(rotation)|t; 0 0 0 1] (inverted for S2C): model.py:790-801, 811-856."""

    @staticmethod
    @_fwd
    def forward(ctx, rotation, translation, invert):
        ctx.invert = invert
        ctx.save_for_backward(rotation, translation)
        return ops.rt_theta(rotation, translation, invert)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dtheta):
        rotation, translation = ctx.saved_tensors
        drot, dtr = ops.rt_theta_bwd(rotation, translation, dtheta.contiguous(), ctx.invert)
        return drot, dtr, None


class UpsampleNearestFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=(sD,sH,sW)), default nearest (model.py:427-433)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, scale):
        ctx.scale = tuple(int(v) for v in scale)
        return ops.upsample_nearest(x.contiguous(), ctx.scale)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dout):
        return ops.upsample_nearest_bwd(dout.contiguous(), ctx.scale), None


class AddMatmulFn(torch.autograd.Function):
    """s = (z + e) @ Gamma (model.py:945-957: right-multiply, no transpose)."""

    @staticmethod
    @_fwd
    def forward(ctx, z, e, gamma):
        ctx.save_for_backward(z, e, gamma)
        return ops.add_matmul(z, e, gamma)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, ds):
        z, e, gamma = ctx.saved_tensors
        ds = ds.contiguous()
        dz = ops.small_gemm(ds, gamma.detach(), trans_b=True) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        dgamma = ops.small_gemm(z, ds, trans_a=True, a2=e) if ctx.needs_input_grad[2] else None
        return dz, dz, dgamma


class Conv1x1OnVectorFn(torch.autograd.Function):
    """The 1x1 Conv2d applied to a 1x1 map (model.py:446): x = s @ W^T + b with W [N,K,1,1]."""

    @staticmethod
    @_fwd
    def forward(ctx, s, weight, bias, w_kn):
        ctx.save_for_backward(s, weight)
        return ops.add_matmul(s, None, w_kn, bias)

    @staticmethod
    @once_differentiable  # no double backward (the reference's losses need none): asking for one raises instead of returning zeros
    @_bwd
    def backward(ctx, dx):
        s, weight = ctx.saved_tensors
        dx = dx.contiguous()
        w2d = weight.detach().reshape(weight.shape[0], weight.shape[1])
        ds = ops.small_gemm(dx, w2d) if ctx.needs_input_grad[0] else None
        dw = ops.small_gemm(dx, s, trans_a=True).view(weight.shape) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            ones = torch.ones((1, dx.shape[0]), dtype=torch.float32, device=dx.device)
            db = ops.small_gemm(ones, dx).view(-1)
        return ds, dw, db, None
