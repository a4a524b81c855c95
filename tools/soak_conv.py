"""Soak of the hand-ordered conv kernels: N launches each of the role-split kernel, the big-tile kernel (MPHIP_WINO_PP=2) and its two-frame
mode, plain and with the fused input GroupNorm, every output compared bitwise with the first (a rare ordering hole in the counted waits
would show as a mismatch).  usage: python tools/soak_conv.py [N=400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(3)
bad = 0
for mode, shape in (("1", (8, 96, 96, 16, 64, 64)), ("2", (8, 96, 96, 16, 64, 64)), ("1", (8, 192, 192, 8, 32, 32)), ("2", (8, 384, 384, 4, 16, 16)),
                    ("1", (8, 768, 768, 2, 8, 8)), ("1", (8, 384, 768, 2, 8, 8)), ("1", (5, 96, 192, 2, 16, 8))):
    os.environ["MPHIP_WINO_PP"] = mode
    n, ci, co, d, h, w = shape
    x = torch.randn(n, ci, d, h, w, device=dev) * 1.3 + 0.2
    pc = ops.PackedConv(torch.randn(co, ci, 3, 3, 3, device=dev) * (ci * 27) ** -0.5, torch.randn(co, device=dev) * 0.1)
    g, be = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.2
    st = ops.groupnorm_stats(x, 32)
    ref = ops.conv3d(x, pc, precision=1).clone()
    refg = ops.conv3d_gn_in(x, st, g, be, 32, pc)
    refg = (refg[0] if isinstance(refg, tuple) else refg).clone()
    miss = 0
    for i in range(N):
        y = ops.conv3d(x, pc, precision=1)
        yg = ops.conv3d_gn_in(x, st, g, be, 32, pc)
        yg = yg[0] if isinstance(yg, tuple) else yg
        if i % 8 == 7 or i == N - 1:
            miss += (not torch.equal(y, ref)) + (not torch.equal(yg, refg))
    bad += miss
    print(f"MPHIP_WINO_PP={mode} {shape}: {N} launches plain + fused, mismatching checks: {miss}", flush=True)
os.environ.pop("MPHIP_WINO_PP", None)
print("soak_conv ok" if bad == 0 else "soak_conv FAILED")
sys.exit(0 if bad == 0 else 1)
