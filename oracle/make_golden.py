"""Generates tests/golden/*.npz by running the REFERENCE ITSELF (imported from /root/reference,
this container only) on seeded inputs/weights.  Only outputs are stored: inputs and weights are
regenerated anywhere from the integer PRNG in oracle/hotpath_ref.py (seeded_*).

    python -m oracle.make_golden            # rewrites tests/golden/

TEST INFRASTRUCTURE.  Fixtures are data (tensors), never reference source.
"""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

from . import hotpath_ref as R
from .import_reference import load_reference_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

WEIGHT_SEED = 7
INPUT_SEED = 3


def _load(mod, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    mod.load_state_dict(sub, strict=True)
    return mod.eval()


def _sha(t: torch.Tensor) -> str:
    return hashlib.sha1(t.contiguous().numpy().tobytes()).hexdigest()


def capture_tables():
    """The host-CPU `torch.linspace` tables the reference's CPU path uses at the hot-path sizes
    (model.py:1040-1042 identity grid; affine_grid base = linspace*(G-1)/G), captured as fp32 bit
    patterns.  ATen's linspace is a vectorised kernel with no closed form and its bits depend on
    the host ISA, so the tables are DATA shared by the product
    (megaportrait-hack_amd/data/) and the oracle — both stay host-independent and pinned to the
    container the goldens were generated in."""
    tabs = {"torch": torch.__version__, "linspace": {}, "affine_base": {}}
    for n in (16, 64):
        tabs["linspace"][str(n)] = torch.linspace(-1, 1, n).numpy().view(np.uint32).tolist()
    for g in (64,):
        tabs["affine_base"][str(g)] = (torch.linspace(-1, 1, g) * (g - 1) / g).numpy().view(np.uint32).tolist()
    path = os.path.join(ROOT, "megaportrait-hack_amd", "data", "linspace_tables.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(tabs, f)
    return path


def make_g2d_head():
    """(10) next-row f3: the entry of G2d (model.py:718-719, 756-757) — the reference's own `reshape` and `conv1x1`
    Conv2d modules applied to a projected feature map [2,96,16,16]."""
    m = load_reference_model()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    g2d = m.G2d(96)
    sd = R.seeded_state_dict(R.g2d_head_shapes(), WEIGHT_SEED + 20, "G2d.")
    g2d.reshape.load_state_dict({"weight": sd["G2d.reshape.weight"], "bias": sd["G2d.reshape.bias"]})
    g2d.conv1x1.load_state_dict({"weight": sd["G2d.conv1x1.weight"], "bias": sd["G2d.conv1x1.bias"]})
    x = R.seeded_tensor((2, 96, 16, 16), 120, scale=2.0)
    with torch.no_grad():
        out = g2d.conv1x1(g2d.reshape(x))
        assert torch.equal(out, R.g2d_head(x, sd))
    np.savez(os.path.join(OUT, "g2d_head.npz"), out=out.numpy())
    print("g2d_head.npz", os.path.getsize(os.path.join(OUT, "g2d_head.npz")))


def make_backward():
    """(11) row f2: gradients autograd produces THROUGH THE REFERENCE'S OWN MODULES (imported), for seeded inputs and
    weights — pins the backward oracle (torch CPU autograd of oracle/hotpath_ref.py) the GPU backward tests compare with.
      * apply_warping_field: d/dv and d/dfield for v [1,8,8,16,16], a wide field
      * WarpGeneratorS2C: d/d(R,t,z,e) and two parameter gradients
      * G3d on 96x8x8x8: d/dx and the gradients of the first and last conv weights / one GroupNorm"""
    m = load_reference_model()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    sd = R.seeded_gbase_hot_state_dict(WEIGHT_SEED)
    out = {}
    # warp
    v = R.seeded_tensor((1, 8, 8, 16, 16), 130, scale=1.7).requires_grad_(True)
    f = ((R.seeded_tensor((1, 3, 64, 64, 64), 131) + 1.0) * torch.tensor([9.0, 9.0, 5.0]).view(1, 3, 1, 1, 1) - 1.5).requires_grad_(True)
    g = R.seeded_tensor((1, 8, 8, 16, 16), 132)
    m.apply_warping_field(v, f).backward(g)
    out["warp_dv"], out["warp_dfield_s2"] = v.grad.numpy(), f.grad[:, :, ::2, ::2, ::2].contiguous().numpy()
    out["warp_dfield_sum"] = f.grad.double().sum(dim=(2, 3, 4)).numpy()
    # generator
    s2c = _load(m.WarpGeneratorS2C(512), sd, "warp_generator_s2c.").train()
    inp = {k: t.clone().requires_grad_(True) for k, t in R.seeded_hot_inputs(1, INPUT_SEED).items() if k in ("Rs", "ts", "zs", "es")}
    w = s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])
    w.backward(R.seeded_tensor(tuple(w.shape), 133))
    for k in inp:
        out["s2c_d" + k] = inp[k].grad.numpy()
    out["s2c_dgamma_s8"] = s2c.adaptive_matrix_gamma.grad[::8, ::8].contiguous().numpy()
    out["s2c_dconv3x3x3"] = s2c.flowfield.conv3x3x3.weight.grad.numpy()
    # G3d
    g3d = _load(m.G3d(96), sd, "G3d.").train()
    x = R.seeded_tensor((1, 96, 8, 8, 8), 134, scale=1.7).requires_grad_(True)
    y = g3d(x)
    y.backward(R.seeded_tensor(tuple(y.shape), 135))
    out["g3d_dx"] = x.grad.numpy()
    out["g3d_dfirst_s4"] = g3d.downsampling[0].conv1.weight.grad[::4, ::4].contiguous().numpy()
    out["g3d_dfinal_s4"] = g3d.final_conv.weight.grad[::4, ::4].contiguous().numpy()
    out["g3d_dgn"] = g3d.downsampling[2].gn1.weight.grad.numpy()
    np.savez(os.path.join(OUT, "backward.npz"), **out)
    print("backward.npz", os.path.getsize(os.path.join(OUT, "backward.npz")))


def main():
    import sys

    if "--only-g2d-head" in sys.argv:
        return make_g2d_head()
    if "--only-backward" in sys.argv:
        return make_backward()
    capture_tables()
    m = load_reference_model()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    sd = R.seeded_gbase_hot_state_dict(WEIGHT_SEED)
    s2c = _load(m.WarpGeneratorS2C(512), sd, "warp_generator_s2c.")
    c2d = _load(m.WarpGeneratorC2D(512), sd, "warp_generator_c2d.")
    g3d = _load(m.G3d(96), sd, "G3d.")
    manifest = {"weight_seed": WEIGHT_SEED, "input_seed": INPUT_SEED, "torch": torch.__version__}

    with torch.no_grad():
        # (1) compute_rt_warp, model.py:777-809 — 8 poses x {invert}, grid 8 full + grid 64 strided
        rot = R.seeded_tensor((8, 3), 101, scale=30.0)
        tr = R.seeded_tensor((8, 3), 102, scale=0.17)
        np.savez(os.path.join(OUT, "rt_warp.npz"),
                 g8=m.compute_rt_warp(rot, tr, invert=False, grid_size=8).numpy(),
                 g8_inv=m.compute_rt_warp(rot, tr, invert=True, grid_size=8).numpy(),
                 g64_s8=m.compute_rt_warp(rot, tr, invert=False, grid_size=64)[:, :, ::8, ::8, ::8].contiguous().numpy(),
                 g64_inv_s8=m.compute_rt_warp(rot, tr, invert=True, grid_size=64)[:, :, ::8, ::8, ::8].contiguous().numpy())

        # (2) FlowField, model.py:439-471 — B=2
        zsum = R.seeded_tensor((2, 512), 103, scale=20.0)
        ff = s2c.flowfield(zsum.unsqueeze(-1).unsqueeze(-1), 0, 0)
        np.savez(os.path.join(OUT, "flowfield.npz"), out=ff.numpy())

        # (3) WarpGeneratorS2C / C2D, model.py:938-975 / 989-1024 — B=1, stride-4 samples
        inp = R.seeded_hot_inputs(1, INPUT_SEED)
        w_s2c = s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])
        w_c2d = c2d(inp["Rd"], inp["td"], inp["zd"], inp["es"])
        np.savez(os.path.join(OUT, "warp_generator.npz"),
                 s2c_s4=w_s2c[:, :, ::4, ::4, ::4].contiguous().numpy(), c2d_s4=w_c2d[:, :, ::4, ::4, ::4].contiguous().numpy(),
                 s2c_sha1=_sha(w_s2c), c2d_sha1=_sha(w_c2d))

        # (4) apply_warping_field, model.py:1028-1065
        #   small: v [1,8,16,16,16] with the S2C field, full output + ramp-derived coordinates
        v_small = R.seeded_tensor((1, 8, 16, 16, 16), 104, scale=1.7)
        out_small = m.apply_warping_field(v_small, w_s2c)
        ramp = torch.zeros(1, 3, 16, 16, 16)
        ramp[:, 0] = torch.arange(16.0).view(1, 1, 1, 16)
        ramp[:, 1] = torch.arange(16.0).view(1, 1, 16, 1)
        ramp[:, 2] = torch.arange(16.0).view(1, 16, 1, 1)
        coords_small = m.apply_warping_field(ramp, w_s2c).permute(0, 2, 3, 4, 1).contiguous()
        #   wide field (covers the whole volume; the faithful field only reaches the 4^3 corner)
        wide = R.seeded_tensor((1, 3, 64, 64, 64), 105, scale=1.0)
        wide = (wide + 1.0) * torch.tensor([9.0, 9.0, 9.0]).view(1, 3, 1, 1, 1) - 2.0
        out_wide = m.apply_warping_field(v_small, wide)
        #   full-size volume: strided samples + per-channel sums + sha1
        vs = inp["vs"]
        vc = m.apply_warping_field(vs, w_s2c)
        np.savez(os.path.join(OUT, "apply_warping_field.npz"), small=out_small.numpy(), coords_small=coords_small.numpy(),
                 wide=out_wide.numpy(), full_s4=vc[:, :, ::2, ::4, ::4].contiguous().numpy(),
                 full_chan_sum=vc.double().sum(dim=(2, 3, 4)).numpy(), full_sha1=_sha(vc))

        # (5) ResBlock3D (96->96, 96->192) on 8^3; ResBlock3D_Adaptive (64->32) on 8^3
        x8 = R.seeded_tensor((1, 96, 8, 8, 8), 106, scale=1.7)
        rb_same = g3d.downsampling[0](x8.clone())
        rb_wide = g3d.downsampling[2](x8.clone())
        x64 = R.seeded_tensor((1, 64, 8, 8, 8), 107, scale=1.7)
        rba = s2c.flowfield.resblock4(x64.clone())
        np.savez(os.path.join(OUT, "resblocks.npz"), rb_96_96=rb_same.numpy(), rb_96_192=rb_wide.numpy(), rba_64_32=rba.numpy())

        # (6) G3d: full output at 96x8x8x8; 96x16x16x16 B=2 strided; full-size strided + stats
        g_small = g3d(x8.clone())
        x16 = R.seeded_tensor((2, 96, 16, 16, 16), 108, scale=1.7)
        g_mid = g3d(x16.clone())
        vc2d = g3d(vc)
        np.savez(os.path.join(OUT, "g3d.npz"), small=g_small.numpy(), mid_s2=g_mid[:, :, ::2, ::2, ::2].contiguous().numpy(),
                 full_s4=vc2d[:, :, ::2, ::4, ::4].contiguous().numpy(), full_chan_mean=vc2d.double().mean(dim=(2, 3, 4)).numpy(),
                 full_chan_absmax=vc2d.abs().amax(dim=(2, 3, 4)).numpy())

        # (7) end-to-end hot slice, model.py:1151-1171, full size [1,96,64,64] + a 16^3 case
        proj = torch.sum(m.apply_warping_field(vc2d, w_c2d), dim=2)
        inp16 = R.seeded_hot_inputs(1, INPUT_SEED + 1, D=16, H=16, W=16)
        w1 = s2c(inp16["Rs"], inp16["ts"], inp16["zs"], inp16["es"])
        w2 = c2d(inp16["Rd"], inp16["td"], inp16["zd"], inp16["es"])
        proj16 = torch.sum(m.apply_warping_field(g3d(m.apply_warping_field(inp16["vs"], w1)), w2), dim=2)
        np.savez(os.path.join(OUT, "hot_slice.npz"), full=proj.numpy(), small16=proj16.numpy())

    # (9) next-row f1: Eapp's 3D tail (model.py:271-290) on a [1,1536,16,16] map (volume 96x16x16x16)
    with torch.no_grad():
        tail_sd = R.seeded_state_dict(R.eapp_tail_shapes(), WEIGHT_SEED + 10, "appearanceEncoder.")
        blocks = {}
        for name in sorted(set(R._EAPP_TAIL_ORDER)):
            blk = m.ResBlock3D_Adaptive(in_channels=96, out_channels=96)
            blk.load_state_dict({k[len("appearanceEncoder." + name + "."):]: v for k, v in tail_sd.items()
                                 if k.startswith("appearanceEncoder." + name + ".")}, strict=True)
            blocks[name] = blk.eval()
        feat = R.seeded_tensor((1, 1536, 16, 16), 110, scale=1.7)
        vs_t = feat.view(1, 96, 16, 16, 16)
        for name in R._EAPP_TAIL_ORDER:
            vs_t = blocks[name](vs_t)
        np.savez(os.path.join(OUT, "eapp_tail.npz"), out_s2=vs_t[:, :, ::2, ::2, ::2].contiguous().numpy(),
                 chan_mean=vs_t.double().mean(dim=(2, 3, 4)).numpy(), sha1=_sha(vs_t))
        assert torch.equal(vs_t, R.eapp_tail3d(feat, tail_sd))

    # (8) state-dict manifest (names/shapes are the checkpoint-layout contract)
    manifest["state_dict"] = {
        "warp_generator_s2c": {k: list(v.shape) for k, v in s2c.state_dict().items()},
        "warp_generator_c2d": {k: list(v.shape) for k, v in c2d.state_dict().items()},
        "G3d": {k: list(v.shape) for k, v in g3d.state_dict().items()},
    }
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    make_g2d_head()
    make_backward()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
