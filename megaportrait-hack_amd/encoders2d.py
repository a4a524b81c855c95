"""The 2D parts of Gbase — Eapp's 2D trunk + global descriptor, Emtn, G2d's body, ImagePyramide — as plain
PyTorch-ROCm modules.

north_star keeps these on PyTorch-ROCm (MIOpen convs); they are NOT part of the HIP hot path.  The reference cannot
construct its own versions offline (torchvision / weight downloads / `.cuda(0)`: SURVEY.md §0 quirk 5), so the
orchestrator (`gbase.Gbase`) needs constructible equivalents to produce an image and an end-to-end number.  These are
this repo's own restatements of the architectures (SURVEY.md Appendix C shapes): same attribute names, parameter
shapes and therefore state-dict keys as the reference (`model.py:54-130,136-173,206-299,600-763,869-907,1070-1085`,
`resnet.py:59-283`), random init, no downloads.  Each 2D module is injectable: pass your own `nn.Module` to
`gbase.Gbase(...)` (e.g. the reference's, built on a host that has torchvision) and it is used instead.

Where the hot path touches them:
* `Eapp` runs its 3D tail (five `ResBlock3D_Adaptive(96,96)`, one applied twice) on the HIP kernels (scope row f1);
* `G2d` enters through the fused 96->512 product of `model.G2dHead` (scope row f3).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import model as M

FEATURE_SIZE_AVG_POOL = 2   # model.py:46
FEATURE_SIZE = (2, 2)       # model.py:47
COMPRESS_DIM = 512          # model.py:48


# ---------------------------------------------------------------------------------------------- Eapp (2D trunk)
class Conv2d_WS(nn.Conv2d):
    """Weight-standardised conv (model.py:54-69): per output filter, subtract the mean and divide by (std + 1e-5)."""

    def forward(self, x):
        w = self.weight
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
        std = w.flatten(1).std(dim=1).view(-1, 1, 1, 1) + 1e-5   # unbiased std, like Tensor.std()
        return F.conv2d(x, w / std, self.bias, self.stride, self.padding, self.dilation, self.groups)


class ResBlock_Custom(nn.Module):
    """model.py:87-129 (dimension=2 only — the 3D variant is unused by Gbase):
    conv_res(x) + conv(relu(gn(conv_ws(relu(gn(x)))))) with parameter-free GroupNorm(32)."""

    def __init__(self, dimension, in_channels, out_channels):
        super().__init__()
        if dimension != 2:
            raise NotImplementedError("ResBlock_Custom: Gbase only builds the 2D variant (model.py:210-212)")
        self.dimension, self.in_channels, self.out_channels = dimension, in_channels, out_channels
        self.conv_res = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv_ws = Conv2d_WS(in_channels, out_channels, 3, padding=1)
        self.conv = nn.Conv2d(out_channels, out_channels, 3, padding=1)

    def forward(self, x):
        skip = self.conv_res(x)
        y = self.conv_ws(F.relu(F.group_norm(x, 32)))
        y = self.conv(F.relu(F.group_norm(y, 32)))
        return y + skip


class _Bottleneck(nn.Module):
    """torchvision-layout ResNet bottleneck (stride on the 3x3), names conv1/bn1/conv2/bn2/conv3/bn3/downsample."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != out:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False), nn.BatchNorm2d(out))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


def _stage(block, inplanes, planes, n, stride):
    layers = [block(inplanes, planes, stride)]
    layers += [block(planes * block.expansion, planes) for _ in range(n - 1)]
    return nn.Sequential(*layers)


class CustomResNet50(nn.Module):
    """model.py:136-173: torchvision ResNet-50 through layer3, adaptive 2x2 pool, 1x1 conv 1024->512."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _stage(_Bottleneck, 64, 64, 3, 1)
        self.layer2 = _stage(_Bottleneck, 256, 128, 4, 2)
        self.layer3 = _stage(_Bottleneck, 512, 256, 6, 2)
        self.adaptive_avg_pool = nn.AdaptiveAvgPool2d(FEATURE_SIZE_AVG_POOL)
        self.conv_reduce = nn.Conv2d(1024, 512, kernel_size=1)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer3(self.layer2(self.layer1(x)))
        return self.conv_reduce(self.adaptive_avg_pool(x))


class Eapp(nn.Module):
    """model.py:206-299.  2D trunk and global descriptor on PyTorch-ROCm; the 3D tail (model.py:276-290) on the HIP
    kernels.  Child order and names are the reference's, including the `resblock3D_96_2`
    double assignment: five 3D blocks exist, one runs twice."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 64, 7, stride=1, padding=3)
        self.resblock_128 = ResBlock_Custom(dimension=2, in_channels=64, out_channels=128)
        self.resblock_256 = ResBlock_Custom(dimension=2, in_channels=128, out_channels=256)
        self.resblock_512 = ResBlock_Custom(dimension=2, in_channels=256, out_channels=512)
        self.resblock3D_96 = M.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_2 = M.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_1 = M.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_1_2 = M.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.resblock3D_96_2_2 = M.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
        self.conv_1 = nn.Conv2d(512, 1536, kernel_size=1)
        self.avgpool = nn.AvgPool2d(kernel_size=2, stride=2)
        self.custom_resnet50 = CustomResNet50()
        self.fc = nn.Linear(2048, COMPRESS_DIM)

    def trunk2d(self, x):
        """image [B,3,H,W] -> conv_1 output [B,1536,H/8,W/8] (model.py:248-268)."""
        out = self.avgpool(self.resblock_128(self.conv(x)))
        out = self.avgpool(self.resblock_256(out))
        out = self.avgpool(self.resblock_512(out))
        return self.conv_1(F.relu(F.group_norm(out, 32)))

    def descriptor(self, x):
        """image -> es [B,512] (model.py:294-298)."""
        return self.fc(torch.flatten(self.custom_resnet50(x), start_dim=1))

    def forward(self, x):
        out = self.trunk2d(x)
        vs = out.view(out.size(0), 96, 16, *out.shape[2:])       # model.py:271 — channel c*16+d is voxel (c, d)
        for name in M.Eapp3DTail._ORDER:                          # model.py:276-290, HIP kernels
            vs = getattr(self, name)(vs)
        return vs, self.descriptor(x)


# ---------------------------------------------------------------------------------------------- Emtn
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + idt)


class CifarResNet18(nn.Module):
    """resnet.py:160-283 as `resnet18(...)` builds it: the CIFAR-style stem (3x3 stride-1 conv, then a max-pool)."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _stage(_BasicBlock, 64, 64, 2, 1)
        self.layer2 = _stage(_BasicBlock, 64, 128, 2, 2)
        self.layer3 = _stage(_BasicBlock, 128, 256, 2, 2)
        self.layer4 = _stage(_BasicBlock, 256, 512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():                                  # resnet.py:211-216
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class _RepVGGDeployBlock(nn.Module):
    def __init__(self, cin, cout, stride, groups):
        super().__init__()
        self.rbr_reparam = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, groups=groups, bias=True)

    def forward(self, x):
        return F.relu(self.rbr_reparam(x))


class SixDRepNetBackbone(nn.Module):
    """6DRepNet head-pose regressor (mysixdrepnet.py:30-69): RepVGG-B1g2 in deploy form (stages of re-parameterised 3x3
    convs + ReLU; blocks [1,4,6,16,1], widths 64/128/256/512/2048, 2 groups on the even layers — mysixdrepnet.py:1215-1289),
    global average pool, Linear(2048, 6), Gram-Schmidt to a rotation matrix (mysixdrepnet.py:272-285)."""

    def __init__(self):
        super().__init__()
        widths, blocks = (128, 256, 512, 2048), (4, 6, 16, 1)
        self.layer0 = _RepVGGDeployBlock(3, 64, 2, 1)
        idx, cin, stages = 1, 64, []
        for w, n in zip(widths, blocks):
            seq = []
            for i in range(n):
                seq.append(_RepVGGDeployBlock(cin, w, 2 if i == 0 else 1, 2 if (idx % 2 == 0 and idx <= 26) else 1))
                cin, idx = w, idx + 1
            stages.append(nn.Sequential(*seq))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.linear_reg = nn.Linear(2048, 6)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.layer0(x)))))
        p = self.linear_reg(torch.flatten(self.gap(x), 1))
        return ortho6d_to_matrix(p[:, :6]), p[:, 6:]


def ortho6d_to_matrix(p):
    """mysixdrepnet.py:272-285 (+ its normalize_vector with the 1e-8 floor)."""
    def unit(v):
        return v / torch.clamp(v.norm(dim=1, keepdim=True), min=1e-8)

    x = unit(p[:, 0:3])
    z = unit(torch.cross(x, p[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack((x, y, z), dim=2)


def euler_from_matrix(R):
    """mysixdrepnet.py:291-315: x, y, z Euler angles (radians) of a batch of rotation matrices."""
    sy = torch.sqrt(R[:, 0, 0] * R[:, 0, 0] + R[:, 1, 0] * R[:, 1, 0])
    sing = (sy < 1e-6).to(R.dtype)
    x = torch.atan2(R[:, 2, 1], R[:, 2, 2]) * (1 - sing) + torch.atan2(-R[:, 1, 2], R[:, 1, 1]) * sing
    y = torch.atan2(-R[:, 2, 0], sy)
    z = torch.atan2(R[:, 1, 0], R[:, 0, 0]) * (1 - sing)
    return torch.stack((x, y, z), dim=1)


class SixDRepNet_Detector:
    """mysixdrepnet.py:770-829.  A plain object, NOT an nn.Module, exactly like the reference's: the pose regressor is a
    frozen pretrained net that is neither trained nor part of Gbase's state-dict (model.py:876).  It follows its input's
    device lazily (the reference calls `.cuda(0)` in the constructor)."""

    def __init__(self, model: nn.Module = None):
        self.model = (model if model is not None else SixDRepNetBackbone()).eval()
        for p in self.model.parameters():
            p.requires_grad_(False)

    def predict(self, img):
        ref = next(self.model.parameters())
        if ref.device != img.device:
            self.model.to(img.device)
        with torch.no_grad():
            rot, trans = self.model(img.float())
        return euler_from_matrix(rot) * 180.0 / math.pi, trans


class Emtn(nn.Module):
    """model.py:869-907: rotation (degrees) from the frozen 6DRepNet, translation = last three outputs of a ResNet-18
    pose head, expression = ResNet-18 features -> fc(2048, 512)."""

    def __init__(self, rotation_net=None):
        super().__init__()
        self.head_pose_net = CifarResNet18(num_classes=1000)
        self.head_pose_net.fc = nn.Linear(512, 6)
        self.rotation_net = rotation_net if rotation_net is not None else SixDRepNet_Detector()
        feat = CifarResNet18(num_classes=512)
        self.expression_net = nn.Sequential(*list(feat.children())[:-1])       # model.py:880: everything but the fc
        self.expression_net.adaptive_pool = nn.AdaptiveAvgPool2d(FEATURE_SIZE)  # model.py:881: appended, runs last
        self.fc = nn.Linear(2048, COMPRESS_DIM)

    def forward(self, x):
        rotations, _ = self.rotation_net.predict(x)
        translation = self.head_pose_net(x)[:, 3:]
        expression = self.fc(torch.flatten(self.expression_net(x), start_dim=1))
        return rotations, translation, expression


# ---------------------------------------------------------------------------------------------- G2d
class ResBlock2D(nn.Module):
    """model.py:600-640 (the `downsample` branch is never enabled by G2d)."""

    def __init__(self, in_channels, out_channels, downsample=False):
        super().__init__()
        self.downsample = downsample
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        if downsample:
            self.downsample_conv = nn.Conv2d(in_channels, out_channels, 1, stride=2)
            self.downsample_bn = nn.BatchNorm2d(out_channels)
        if in_channels != out_channels:
            self.shortcut = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride=1), nn.BatchNorm2d(out_channels))
        else:
            self.shortcut = nn.Identity()

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        idt = self.downsample_bn(self.downsample_conv(x)) if self.downsample else x
        return F.relu(y + self.shortcut(idt))


class G2d(M.G2dHead):
    """model.py:715-763.  `reshape` + `conv1x1` are inherited from model.G2dHead (scope row f3: one fused 96->512
    product on the HIP kernels at inference); the body — 8 ResBlock2D(512), three bilinear x2 + ResBlock2D stages,
    GN-ReLU-conv-sigmoid — is PyTorch-ROCm."""

    def __init__(self, in_channels=96):
        super().__init__()
        self.res_blocks = nn.Sequential(*[ResBlock2D(512, 512) for _ in range(8)])
        up = lambda: nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.upsample1 = nn.Sequential(up(), ResBlock2D(512, 256))
        self.upsample2 = nn.Sequential(up(), ResBlock2D(256, 128))
        self.upsample3 = nn.Sequential(up(), ResBlock2D(128, 64))
        self.final_conv = nn.Sequential(nn.GroupNorm(32, 64), nn.ReLU(inplace=True), nn.Conv2d(64, 3, 3, padding=1), nn.Sigmoid())

    def body(self, x):
        """[B,512,h,w] (the head's output) -> image [B,3,8h,8w] in (0,1) (model.py:758-762)."""
        x = self.res_blocks(x)
        x = self.upsample3(self.upsample2(self.upsample1(x)))
        return self.final_conv(x)

    def forward(self, x):
        return self.body(M.G2dHead.forward(self, x))


# ---------------------------------------------------------------------------------------------- ImagePyramide
class AntiAliasInterpolation2d(nn.Module):
    """model.py:646-691: gaussian blur (sigma = (1/scale - 1)/2, kernel 2*round(4 sigma)+1) then nearest sub-sampling;
    the kernel is the buffer `weight` [C,1,k,k] (in the state-dict)."""

    def __init__(self, channels, scale):
        super().__init__()
        sigma = (1 / scale - 1) / 2
        k = 2 * round(sigma * 4) + 1
        self.ka = k // 2
        self.kb = self.ka - 1 if k % 2 == 0 else self.ka
        ax = torch.arange(k, dtype=torch.float32)
        g = torch.exp(-(ax - (k - 1) / 2) ** 2 / (2 * sigma ** 2))
        kernel = g[:, None] * g[None, :]
        kernel = kernel / kernel.sum()
        self.register_buffer("weight", kernel.view(1, 1, k, k).repeat(channels, 1, 1, 1))
        self.groups, self.scale = channels, scale

    def forward(self, x):
        if self.scale == 1.0:
            return x
        out = F.conv2d(F.pad(x, (self.ka, self.kb, self.ka, self.kb)), weight=self.weight, groups=self.groups)
        return F.interpolate(out, scale_factor=(self.scale, self.scale))


class ImagePyramide(nn.Module):
    """model.py:1070-1085: {'prediction_<scale>': blurred + sub-sampled image} for the pyramid perceptual loss."""

    def __init__(self, scales, num_channels):
        super().__init__()
        self.downs = nn.ModuleDict({str(s).replace(".", "-"): AntiAliasInterpolation2d(num_channels, s) for s in scales})

    def forward(self, x):
        return {"prediction_" + name.replace("-", "."): down(x) for name, down in self.downs.items()}
