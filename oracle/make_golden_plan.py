"""Fixture for the plain-C test of the one-call entry (tests/c_abi/plan_smoke.c): a MANIFEST of the integer-PRNG tensors
(state-dict name, element count, LCG seed, scale, shift — data, no code) and the expected output of the hot slice on them,
computed by the CPU oracle (oracle/hotpath_ref.py, pinned to the imported reference by tests/test_oracle.py).

    python oracle/make_golden_plan.py      # writes tests/golden/plan_c_manifest.txt and tests/golden/plan_c_expected.bin

The C program regenerates every tensor from the manifest with the same 64-bit LCG; this script checks that a float32
emulation of that generator is bit-identical to the torch tensors the oracle consumed before it writes anything."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hotpath_ref as R  # noqa: E402

B, D, H, W = 1, 16, 16, 16
SEED_W, SEED_IN = 57, 6


def lcg_f32(n, seed, scale, shift):
    """What plan_smoke.c does: state = a*state + c (mod 2^64), u = top 24 bits / 2^24, v = float(2u-1) * float(scale) + float(shift)."""
    a, c, mask = 6364136223846793005, 1442695040888963407, (1 << 64) - 1
    st = (seed * 2654435761 + 88172645463325252) & mask
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        st = (a * st + c) & mask
        out[i] = np.float32((st >> 40) / float(1 << 24) * 2.0 - 1.0)
    return out * np.float32(scale) + np.float32(shift)


def entries():
    """(name, shape, seed, scale, shift) exactly as hotpath_ref.seeded_gbase_hot_state_dict / seeded_hot_inputs draw them."""
    out = []
    for prefix, shapes, seed in (("warp_generator_s2c.", R.warp_generator_shapes(), SEED_W + 1), ("warp_generator_c2d.", R.warp_generator_shapes(), SEED_W + 2),
                                 ("G3d.", R.g3d_shapes(), SEED_W + 3)):
        for i, (k, shp) in enumerate(sorted(shapes.items())):
            s = seed * 1000 + i
            wk = k.replace("bias", "weight")
            if "adaptive_matrix" in k:
                scale, shift = math.sqrt(3.0), 0.0
            elif k.endswith("weight") and len(shp) >= 4 and shp[0] != 1:
                scale, shift = 1.0 / math.sqrt(int(math.prod(shp[1:]))), 0.0
            elif k.endswith("bias") and wk in shapes and len(shapes[wk]) >= 4 and shapes[wk][0] != 1:
                scale, shift = 1.0 / math.sqrt(int(math.prod(shapes[wk][1:]))), 0.0
            elif k.endswith("weight"):
                scale, shift = 0.25, 1.0
            else:
                scale, shift = 0.25, 0.0
            out.append((prefix + k, shp, s, scale, shift))
    r3 = math.sqrt(3.0)
    for j, (name, shp, scale) in enumerate((("vs", (B, 96, D, H, W), r3), ("es", (B, 512), r3), ("zs", (B, 512), r3), ("zd", (B, 512), r3),
                                            ("Rs", (B, 3), 30.0), ("Rd", (B, 3), 30.0), ("ts", (B, 3), 0.17), ("td", (B, 3), 0.17)), start=1):
        out.append(("input." + name, shp, SEED_IN * 100 + j, scale, 0.0))
    return out


def main():
    sd = R.seeded_gbase_hot_state_dict(SEED_W)
    inp = R.seeded_hot_inputs(B, SEED_IN, D=D, H=H, W=W)
    lines = []
    for name, shp, seed, scale, shift in entries():
        want = (inp[name[len("input."):]] if name.startswith("input.") else sd[name]).reshape(-1).numpy()
        n = want.size
        if n <= 70000 or name.endswith("final_conv.weight"):   # the emulation is a Python loop: check every small tensor and one large one
            got = lcg_f32(n, seed, scale, shift)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"LCG emulation differs from torch for {name}"
        lines.append(f"{name} {n} {seed} {np.float32(scale).view(np.uint32):08x} {np.float32(shift).view(np.uint32):08x}")
    with torch.no_grad():
        out = R.hot_slice(sd=sd, **inp)
    assert out.shape == (B, 96, H, W)
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "plan_c_manifest.txt"), "w") as f:
        f.write(f"# name numel lcg_seed scale_bits shift_bits  (oracle/make_golden_plan.py; volume {B}x96x{D}x{H}x{W})\n")
        f.write("\n".join(lines) + "\n")
    out.numpy().astype("<f4").tofile(os.path.join(gold, "plan_c_expected.bin"))
    print(f"wrote {len(lines)} manifest entries and {out.numel()} expected floats (|out|max {out.abs().max():.3f})")


if __name__ == "__main__":
    main()
