"""Golden vectors for the orchestrator row (SURVEY.md §8b `Gbase`) — run ONLY in the build container, where
/root/reference exists:  python -m oracle.make_golden_gbase

Test infrastructure (never imported by the product).  The 2D parts of Gbase stay PyTorch-ROCm (north_star); this
repo's constructible restatements of them (megaportrait-hack_amd/encoders2d.py) are pinned here against the
REFERENCE's own modules on integer-PRNG weights/inputs at small sizes.  Only tensors and name/shape manifests are
stored (tests/golden/gbase2d.npz, gbase_manifest.json) — no reference source.

Reference modules exercised: ResBlock_Custom / Conv2d_WS (model.py:54-129), Eapp's 2D trunk call order
(model.py:248-268), ResBlock2D + G2d body (model.py:600-640, 758-762), ImagePyramide (model.py:646-691, 1070-1085),
resnet18 (resnet.py:160-304), MySixDRepNet + 6D->matrix->Euler (mysixdrepnet.py:30-69, 272-315).
"""
import importlib
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import hotpath_ref as R  # noqa: E402
from oracle.import_reference import REFERENCE_ROOT, load_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 500


def seeded_module_state(module: torch.nn.Module, seed: int):
    """Integer-PRNG values for every entry of a module's state-dict (BatchNorm statistics included: running_var > 0)."""
    sd = {}
    for i, (k, v) in enumerate(sorted(module.state_dict().items())):
        s = seed * 1000 + i
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            t = torch.zeros(shp, dtype=v.dtype)
        elif k.endswith("running_var"):
            t = R.seeded_tensor(shp, s, scale=0.4, shift=1.0)
        elif k.endswith("running_mean"):
            t = R.seeded_tensor(shp, s, scale=0.2)
        elif k.endswith("weight") and len(shp) >= 2:
            t = R.seeded_tensor(shp, s, scale=1.0 / math.sqrt(max(1, int(np.prod(shp[1:])))))
        elif k.endswith("weight"):
            t = R.seeded_tensor(shp, s, scale=0.25, shift=1.0)
        else:
            t = R.seeded_tensor(shp, s, scale=0.1)
        sd[k] = t
    return sd


def _load(module, seed):
    module.load_state_dict(seeded_module_state(module, seed), strict=True)
    return module.eval()


def _ref_submodule(name):
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        return importlib.import_module(name)
    finally:
        sys.path.remove(REFERENCE_ROOT)
        sys.modules.pop(name, None)


def main():
    m = load_reference_model()
    out, manifest = {}, {}
    with torch.no_grad():
        # (1) ResBlock_Custom(2, 64, 128), model.py:87-129
        blk = _load(m.ResBlock_Custom(dimension=2, in_channels=64, out_channels=128), SEED + 1)
        out["rbc_64_128"] = blk(R.seeded_tensor((1, 64, 16, 16), SEED + 2, scale=1.7)).numpy()

        # (2) Eapp's 2D trunk on a 64x64 image (model.py:248-268): conv -> RB128 -> pool -> RB256 -> pool -> RB512 -> pool
        #     -> group_norm(32) -> relu -> conv_1.  Eapp() constructs here with the torchvision stub (its resnet50 pieces
        #     are inert mocks); only the trunk modules are used.
        eapp = m.Eapp()
        trunk_keys = ("conv", "resblock_128", "resblock_256", "resblock_512", "conv_1")
        for i, name in enumerate(trunk_keys):
            _load(getattr(eapp, name), SEED + 10 + i)
        img = (R.seeded_tensor((1, 3, 64, 64), SEED + 20) + 1.0) * 0.5
        t = eapp.avgpool(eapp.resblock_128(eapp.conv(img)))
        t = eapp.avgpool(eapp.resblock_256(t))
        t = eapp.avgpool(eapp.resblock_512(t))
        out["eapp_trunk_64"] = eapp.conv_1(F.relu(F.group_norm(t, num_groups=32))).numpy()
        manifest["appearanceEncoder"] = {k: list(v.shape) for k, v in eapp.state_dict().items()}

        # (3) ResBlock2D(512, 256) (model.py:600-640), eval mode with non-trivial BatchNorm statistics
        rb2 = _load(m.ResBlock2D(512, 256), SEED + 30)
        out["rb2d_512_256"] = rb2(R.seeded_tensor((1, 512, 8, 8), SEED + 31, scale=1.7)).numpy()

        # (4) G2d (model.py:715-763): whole module on a [1,96,4,4] projection (head + body), and the body alone
        g2d = _load(m.G2d(96), SEED + 40)
        p = R.seeded_tensor((1, 96, 4, 4), SEED + 41, scale=2.0)
        out["g2d_full_4"] = g2d(p).numpy()
        h = g2d.conv1x1(g2d.reshape(p))
        out["g2d_head_4"] = h.numpy()
        manifest["G2d"] = {k: list(v.shape) for k, v in g2d.state_dict().items()}

        # (5) ImagePyramide (model.py:1070-1085)
        pyr = m.ImagePyramide(scales=[0.5, 0.25], num_channels=3)
        res = pyr(img)
        for k, v in res.items():
            out["pyr_" + k] = v.numpy()
        manifest["image_pyramid"] = {k: list(v.shape) for k, v in pyr.state_dict().items()}

        # (6) resnet.py resnet18 (CIFAR stem) as Emtn uses it (model.py:873-881)
        resnet = _ref_submodule("resnet")
        r18 = _load(resnet.resnet18(pretrained=False, num_classes=512), SEED + 50)
        x32 = (R.seeded_tensor((2, 3, 32, 32), SEED + 51) + 1.0) * 0.5
        out["r18_logits"] = r18(x32).numpy()
        expr = torch.nn.Sequential(*list(r18.children())[:-1])
        expr.adaptive_pool = torch.nn.AdaptiveAvgPool2d((2, 2))
        out["r18_expression_feat"] = torch.flatten(expr(x32), start_dim=1).numpy()
        manifest["motionEncoder.expression_net"] = {k: list(v.shape) for k, v in expr.state_dict().items()}
        hp = resnet.resnet18(pretrained=False)
        hp.fc = torch.nn.Linear(hp.fc.in_features, 6)
        manifest["motionEncoder.head_pose_net"] = {k: list(v.shape) for k, v in hp.state_dict().items()}
        manifest["motionEncoder.fc"] = {"weight": [512, 2048], "bias": [512]}

        # (7) 6DRepNet: RepVGG-B1g2 deploy -> 6D -> matrix -> Euler degrees (mysixdrepnet.py:30-69, 272-315, 810-818)
        six = _ref_submodule("mysixdrepnet")
        net = _load(six.MySixDRepNet(backbone_name="RepVGG-B1g2", backbone_file="", deploy=True, pretrained=False), SEED + 60)
        x64 = (R.seeded_tensor((2, 3, 64, 64), SEED + 61) + 1.0) * 0.5
        rot, _ = net(x64)
        out["six_rotmat"] = rot.numpy()
        out["six_euler_deg"] = (six.compute_euler_angles_from_rotation_matrices(rot) * 180 / np.pi).numpy()
        p6 = R.seeded_tensor((16, 6), SEED + 62, scale=2.0)
        r6 = six.compute_rotation_matrix_from_ortho6d(p6)
        out["ortho6d_matrix"] = r6.numpy()
        out["ortho6d_euler_rad"] = six.compute_euler_angles_from_rotation_matrices(r6).numpy()
        manifest["rotation_net.model"] = {k: list(v.shape) for k, v in net.state_dict().items()}

    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "gbase2d.npz"), **out)
    with open(os.path.join(OUT, "gbase_manifest.json"), "w") as f:
        json.dump({"seed": SEED, "torch": torch.__version__, "state_dict": manifest}, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()))
    print({k: len(v) for k, v in manifest.items()})


if __name__ == "__main__":
    main()
