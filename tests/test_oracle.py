"""CPU tests of the oracle itself (no GPU): the two restatements (ATen-functional hotpath_ref.py and
plain-C hotpath_c.c) against each other, against the golden fixtures produced by the reference
(tests/golden/, oracle/make_golden.py) and — in the build container only — against the imported
reference.  This is what "pins" the oracle."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hotpath_ref as R
from oracle.import_reference import reference_available

GOLD = os.path.join(os.path.dirname(__file__), "golden")
WEIGHT_SEED, INPUT_SEED = 7, 3


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def sd():
    return R.seeded_gbase_hot_state_dict(WEIGHT_SEED)


def maxabs(a, b):
    return (torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max().item()


def test_prng_is_stable():
    t = R.seeded_tensor((5,), 42)
    assert t.dtype == torch.float32
    # pinned values: the integer PRNG must generate the same inputs on every host
    assert np.array_equal(t.numpy().view(np.uint32), R.seeded_tensor((5,), 42).numpy().view(np.uint32))
    assert abs(float(R.seeded_tensor((100000,), 1).mean())) < 0.01
    assert R.seeded_tensor((3, 4), 9)[1, 2] == R.seeded_tensor((12,), 9)[6]


def test_state_dict_manifest(sd):
    """Key names / shapes of the seeded state-dict == those of the reference modules."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["state_dict"]
    for prefix in ("warp_generator_s2c", "warp_generator_c2d", "G3d"):
        ours = {k[len(prefix) + 1:]: list(v.shape) for k, v in sd.items() if k.startswith(prefix + ".")}
        assert ours == man[prefix]


# ------------------------------------------------------------------ ATen restatement vs reference goldens
def test_ref_rt_warp_golden():
    g = gold("rt_warp")
    rot, tr = R.seeded_tensor((8, 3), 101, scale=30.0), R.seeded_tensor((8, 3), 102, scale=0.17)
    assert maxabs(R.compute_rt_warp(rot, tr, False, 8), g["g8"]) < 1e-6
    assert maxabs(R.compute_rt_warp(rot, tr, True, 8), g["g8_inv"]) < 1e-6
    assert maxabs(R.compute_rt_warp(rot, tr, True, 64)[:, :, ::8, ::8, ::8], g["g64_inv_s8"]) < 1e-6


def test_ref_flowfield_and_generators_golden(sd):
    zsum = R.seeded_tensor((2, 512), 103, scale=20.0)
    assert maxabs(R.flowfield(zsum, sd, "warp_generator_s2c.flowfield."), gold("flowfield")["out"]) < 1e-5
    inp = R.seeded_hot_inputs(1, INPUT_SEED)
    g = gold("warp_generator")
    w1 = R.warp_generator(inp["Rs"], inp["ts"], inp["zs"], inp["es"], sd, "warp_generator_s2c.", True)
    w2 = R.warp_generator(inp["Rd"], inp["td"], inp["zd"], inp["es"], sd, "warp_generator_c2d.", False)
    assert maxabs(w1[:, :, ::4, ::4, ::4], g["s2c_s4"]) < 1e-5
    assert maxabs(w2[:, :, ::4, ::4, ::4], g["c2d_s4"]) < 1e-5


def test_ref_resblocks_g3d_golden(sd):
    g = gold("resblocks")
    x8 = R.seeded_tensor((1, 96, 8, 8, 8), 106, scale=1.7)
    assert maxabs(R.resblock3d(x8, sd, "G3d.downsampling.0."), g["rb_96_96"]) < 1e-5
    assert maxabs(R.resblock3d(x8, sd, "G3d.downsampling.2."), g["rb_96_192"]) < 1e-5
    x64 = R.seeded_tensor((1, 64, 8, 8, 8), 107, scale=1.7)
    assert maxabs(R.resblock3d_adaptive(x64, sd, "warp_generator_s2c.flowfield.resblock4."), g["rba_64_32"]) < 1e-5
    assert maxabs(R.g3d(x8, sd), gold("g3d")["small"]) < 1e-5


def test_ref_hot_slice_small_golden(sd):
    inp = R.seeded_hot_inputs(1, INPUT_SEED + 1, D=16, H=16, W=16)
    assert maxabs(R.hot_slice(sd=sd, **inp), gold("hot_slice")["small16"]) < 1e-4


def test_ref_eapp_tail_golden():
    """Next-row f1 (Eapp 3D tail, model.py:271-290): five blocks, six applications."""
    sd_t = R.seeded_state_dict(R.eapp_tail_shapes(), WEIGHT_SEED + 10, "appearanceEncoder.")
    assert len({k.split(".")[1] for k in sd_t}) == 5
    feat = R.seeded_tensor((1, 1536, 16, 16), 110, scale=1.7)
    out = R.eapp_tail3d(feat, sd_t)
    g = gold("eapp_tail")
    assert out.shape == (1, 96, 16, 16, 16)
    assert maxabs(out[:, :, ::2, ::2, ::2], g["out_s2"]) < 1e-5


# ------------------------------------------------------------------ plain-C restatement
def test_c_index_pipeline_matches_aten_here(oracle_c):
    """Bit-level pin of the C oracle against ATen's CPU kernels.  ATen's FMA use depends on the host
    ISA, so the bitwise assertion is made where the goldens were generated (reference present =
    build container); elsewhere a 1e-6 tolerance."""
    exact = reference_available()
    theta = R.seeded_tensor((3, 3, 4), 201)
    em = (R.seeded_tensor((3, 3, 16, 16, 16), 202) + 1.0) * 0.5
    pairs = [
        (oracle_c.affine_grid3d(theta, 64), F.affine_grid(theta, (3, 1, 64, 64, 64), align_corners=False).permute(0, 4, 1, 2, 3)),
        (oracle_c.resize_trilinear(em, (64, 64, 64), False), F.interpolate(em, size=(64, 64, 64), mode="trilinear", align_corners=False)),
        (oracle_c.resize_trilinear(em, (32, 32, 32), True), F.interpolate(em, scale_factor=2, mode="trilinear", align_corners=True)),
        (oracle_c.avgpool2(em), F.avg_pool3d(em, 2, 2)),
        (oracle_c.upsample_nearest(em, (1, 2, 2)), F.interpolate(em, scale_factor=(1, 2, 2), mode="nearest")),
    ]
    for got, want in pairs:
        assert torch.equal(got, want.contiguous()) if exact else maxabs(got, want) < 1e-6
    for field in ((R.seeded_tensor((2, 3, 64, 64, 64), 301, scale=1.3) + 0.4),
                  (R.seeded_tensor((2, 3, 64, 64, 64), 302) + 1.0) * torch.tensor([34.0, 34.0, 9.0]).view(1, 3, 1, 1, 1) - 2.0):
        c, i = oracle_c.warp_coords(field, 16, 64, 64)
        c2, i2 = R.warp_coords(field, 16, 64, 64)
        v = R.seeded_tensor((2, 4, 16, 64, 64), 310, scale=1.7)
        if exact:
            assert torch.equal(c, c2) and torch.equal(i, i2)
            assert torch.equal(oracle_c.apply_warping_field(v, field), R.apply_warping_field(v, field))
        else:
            assert maxabs(c, c2) < 1e-4
        assert maxabs(oracle_c.apply_warping_field(v, field, dsum=True), R.apply_warping_field(v, field).sum(2)) < 1e-4


def test_c_coords_golden(oracle_c, sd):
    """Reference-derived coordinates (three-ramp trick through the reference's apply_warping_field)."""
    inp = R.seeded_hot_inputs(1, INPUT_SEED)
    w_s2c = oracle_c.warp_generator(inp["Rs"], inp["ts"], inp["zs"], inp["es"], sd, "warp_generator_s2c.", True)
    g = gold("warp_generator")
    assert maxabs(w_s2c[:, :, ::4, ::4, ::4], g["s2c_s4"]) < 1e-5
    c, idx = oracle_c.warp_coords(w_s2c, 16, 16, 16)
    ga = gold("apply_warping_field")
    assert maxabs(c, ga["coords_small"]) < 1e-4
    assert idx.min() >= 0 and idx[..., 0].max() <= 15
    v_small = R.seeded_tensor((1, 8, 16, 16, 16), 104, scale=1.7)
    assert maxabs(oracle_c.apply_warping_field(v_small, w_s2c), ga["small"]) < 1e-4


def test_c_blocks_vs_golden(oracle_c, sd):
    g = gold("resblocks")
    x64 = R.seeded_tensor((1, 64, 8, 8, 8), 107, scale=1.7)
    assert maxabs(oracle_c.resblock3d_adaptive(x64, sd, "warp_generator_s2c.flowfield.resblock4."), g["rba_64_32"]) < 1e-4
    x = R.seeded_tensor((1, 96, 2, 4, 4), 109, scale=1.7)   # small: the C conv is a naive loop
    assert maxabs(oracle_c.resblock3d(x, sd, "G3d.downsampling.0."), R.resblock3d(x, sd, "G3d.downsampling.0.")) < 1e-4
    zsum = R.seeded_tensor((1, 512), 103, scale=20.0)
    assert maxabs(oracle_c.flowfield(zsum, sd, "warp_generator_c2d.flowfield."), R.flowfield(zsum, sd, "warp_generator_c2d.flowfield.")) < 1e-4


# ------------------------------------------------------------------ against the live reference (container only)
@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_restatement_equals_reference_modules(sd):
    from oracle.import_reference import load_reference_model

    m = load_reference_model()
    s2c, g3d = m.WarpGeneratorS2C(512), m.G3d(96)
    s2c.load_state_dict({k[len("warp_generator_s2c."):]: v for k, v in sd.items() if k.startswith("warp_generator_s2c.")})
    g3d.load_state_dict({k[len("G3d."):]: v for k, v in sd.items() if k.startswith("G3d.")})
    inp = R.seeded_hot_inputs(1, 5, D=8, H=8, W=8)
    with torch.no_grad():
        w = s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])
        assert torch.equal(w, R.warp_generator(inp["Rs"], inp["ts"], inp["zs"], inp["es"], sd, "warp_generator_s2c.", True))
        vc = m.apply_warping_field(inp["vs"], w)
        assert torch.equal(vc, R.apply_warping_field(inp["vs"], w))
        assert torch.equal(g3d(vc.clone()), R.g3d(vc, sd))


def _backward_cases(sd):
    """(name, gradient tensor) pairs of the oracle's autograd for the cases of tests/golden/backward.npz
    (oracle/make_golden.py::make_backward ran the same inputs through the REFERENCE's modules)."""
    out = {}
    v = R.seeded_tensor((1, 8, 8, 16, 16), 130, scale=1.7).requires_grad_(True)
    f = ((R.seeded_tensor((1, 3, 64, 64, 64), 131) + 1.0) * torch.tensor([9.0, 9.0, 5.0]).view(1, 3, 1, 1, 1) - 1.5).requires_grad_(True)
    R.apply_warping_field(v, f).backward(R.seeded_tensor((1, 8, 8, 16, 16), 132))
    out["warp_dv"], out["warp_dfield_s2"] = v.grad, f.grad[:, :, ::2, ::2, ::2]
    out["warp_dfield_sum"] = f.grad.double().sum(dim=(2, 3, 4))
    inp = {k: t.clone().requires_grad_(True) for k, t in R.seeded_hot_inputs(1, INPUT_SEED).items() if k in ("Rs", "ts", "zs", "es")}
    p = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    w = R.warp_generator(inp["Rs"], inp["ts"], inp["zs"], inp["es"], p, "warp_generator_s2c.", invert=True)
    w.backward(R.seeded_tensor(tuple(w.shape), 133))
    for k in inp:
        out["s2c_d" + k] = inp[k].grad
    out["s2c_dgamma_s8"] = p["warp_generator_s2c.adaptive_matrix_gamma"].grad[::8, ::8]
    out["s2c_dconv3x3x3"] = p["warp_generator_s2c.flowfield.conv3x3x3.weight"].grad
    x = R.seeded_tensor((1, 96, 8, 8, 8), 134, scale=1.7).requires_grad_(True)
    p = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    y = R.g3d(x, p)
    y.backward(R.seeded_tensor(tuple(y.shape), 135))
    out["g3d_dx"] = x.grad
    out["g3d_dfirst_s4"] = p["G3d.downsampling.0.conv1.weight"].grad[::4, ::4]
    out["g3d_dfinal_s4"] = p["G3d.final_conv.weight"].grad[::4, ::4]
    out["g3d_dgn"] = p["G3d.downsampling.2.gn1.weight"].grad
    return out


def test_backward_oracle_pinned_by_reference_gradients(sd):
    """The backward oracle (torch CPU autograd of this restatement) reproduces the gradients autograd computed through
    the reference's own modules (tests/golden/backward.npz)."""
    g = np.load(os.path.join(GOLD, "backward.npz"))
    for name, t in _backward_cases(sd).items():
        want = torch.as_tensor(g[name]).double()
        err = (t.detach().double() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert err < 1e-5, (name, err)


def test_winograd_f16_contract_oracle_is_the_conv_when_nothing_rounds():
    """oracle.hotpath_ref.conv3d_wino_f16_contract (the gate of the autocast-mode parity tests) pinned against ATen's conv3d:
    on operands whose transformed values are exactly representable in f16 (small integers; weights that are multiples of 2) nothing is
    rounded anywhere, so the contract must EQUAL the float64 conv — which fixes its transforms, tap order, pairing along W, padding and
    unscale; on real-valued operands it must sit at f16-rounding distance from the truth (not closer: it does round; not farther: one
    rounding per operand), and differ from the direct-domain f16 contract (it rounds t = Bt x and u = G g, not x and g)."""
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-8, 9, (2, 5, 3, 4, 6), generator=g).float()
    w = 2.0 * torch.randint(-4, 5, (7, 5, 3, 3, 3), generator=g).float()
    b = torch.randint(-3, 4, (7,), generator=g).float()
    assert torch.equal(R.conv3d_wino_f16_contract(x, w, b), F.conv3d(x.double(), w.double(), b.double(), padding=1))
    x = R.seeded_tensor((1, 16, 4, 8, 8), 901, scale=1.7)
    w = R.seeded_tensor((24, 16, 3, 3, 3), 902, scale=(16 * 27) ** -0.5)
    truth = F.conv3d(x.double(), w.double(), None, padding=1)
    top = truth.abs().max().item()
    e_contract = (R.conv3d_wino_f16_contract(x, w) - truth).abs().max().item() / top
    e_direct = (R.conv3d_f16_operands(x, w, None, padding=1) - truth).abs().max().item() / top
    assert 2e-5 < e_contract < 3e-3 and 2e-5 < e_direct < 3e-3, (e_contract, e_direct)
    assert (R.conv3d_wino_f16_contract(x, w) - R.conv3d_f16_operands(x, w, None, padding=1)).abs().max().item() / top > 2e-5
    # operand scales: the kernels' rules (max|x| * s in [2^13, 2^14) for activations, [2^14, 2^15) for weights), powers of two
    for m in (1e-8, 0.3, 1.0, 7.5, 3e4):
        for top_bit in (14, 15):
            s = R.pow2_operand_scale(m, top_bit)
            assert 2.0 ** (top_bit - 1) <= m * s < 2.0 ** top_bit and math.log2(s) == round(math.log2(s))
    tiny = R.seeded_tensor((64,), 903) * 1e-7                      # unscaled .half() would flush these into subnormals
    r = R.round_f16_at_scale(tiny)
    assert ((r - tiny.double()).abs() <= 2.0 ** -11 * tiny.double().abs() + 1e-30).all()
