"""dev: per-layer sweep of the f16x3 planner's choices (tile variant x split-K) on G3d's level 1-3 convs: sweep_conv_plans.py [B ...]
One process: the planner reads MPHIP_F16X3_TILE / MPHIP_F16X3_SPLITS on every call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from megaportrait_hack_amd import ops, _lib
LAYERS = [  # name, Ci, Co, D, H, W
    ("L1 b1.conv1", 96, 192, 8, 32, 32), ("L1 b1.conv2", 192, 192, 8, 32, 32), ("L1 u4.conv1", 192, 96, 8, 32, 32), ("L1 u4.conv2", 96, 96, 8, 32, 32),
    ("L2 b2.conv1", 192, 384, 4, 16, 16), ("L2 b2.conv2", 384, 384, 4, 16, 16), ("L2 u2.conv1", 384, 192, 4, 16, 16), ("L2 u2.conv2", 192, 192, 4, 16, 16),
    ("L3 b3.conv1", 384, 768, 2, 8, 8), ("L3 b3.conv2", 768, 768, 2, 8, 8), ("L3 u0.conv1", 768, 384, 2, 8, 8), ("L3 u0.conv2", 384, 384, 2, 8, 8)]
_lib.load()
dev = torch.device("cuda:0")


def time_one(x, pc):
    try:
        for _ in range(3): ops.conv3d(x, pc, precision=1)
    except Exception:
        return float("nan")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv3d(x, pc, precision=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for B in [int(a) for a in sys.argv[1:]] or [8]:
    print(f"--- B={B} (us per conv incl. split-K reduce; auto = the planner's choice)")
    for name, ci, co, d, h, w in LAYERS:
        x = torch.randn(B, ci, d, h, w, device=dev)
        pc = ops.PackedConv(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02, torch.randn(co, device=dev))
        os.environ.pop("MPHIP_F16X3_TILE", None); os.environ.pop("MPHIP_F16X3_SPLITS", None)
        time_one(x, pc)
        base = time_one(x, pc)
        res = []
        for tile in ("0",):   # (the 512-voxel tile variant "1" was removed in r05)
            for sp in ("1", "2", "3", "4", "6", "8", "12", "16", "24", "48"):
                os.environ["MPHIP_F16X3_TILE"] = tile
                os.environ["MPHIP_F16X3_SPLITS"] = sp
                res.append((time_one(x, pc), f"t{tile}s{sp}"))
        os.environ.pop("MPHIP_F16X3_TILE", None); os.environ.pop("MPHIP_F16X3_SPLITS", None)
        best = min(r for r in res if r[0] == r[0])
        gf = 2.0 * B * d * h * w * ci * co * 27 / 1e9
        print(f"{name:14s} {gf:6.1f} GF  auto {base:7.1f} ({gf / base * 1e3:5.0f} TF/s)  best {best[0]:7.1f} [{best[1]}]  " +
              " ".join(f"{n}:{u:.0f}" for u, n in res), flush=True)
