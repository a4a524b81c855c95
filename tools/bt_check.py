"""Dev check of the big-tile F(2,3) conv (conv3d_f16x3_wino_bt.hip, MPHIP_WINO_PP=2) against the role-split kernel (MPHIP_WINO_PP=1):
BIT-IDENTICAL outputs and GroupNorm statistics expected (same arithmetic, same accumulation order), plain / fused-GroupNorm input /
split-K / demand-driven, then interleaved timings.  usage: python tools/bt_check.py [--time-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"


def run(fn, mode):
    os.environ["MPHIP_WINO_PP"] = str(mode)
    try:
        return fn()
    finally:
        os.environ.pop("MPHIP_WINO_PP", None)


def check(N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W) * 1.7
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    yo, so = run(lambda: ops.conv3d(xd, pc, precision=1, gn_groups=32), 1)
    yn, sn = run(lambda: ops.conv3d(xd, pc, precision=1, gn_groups=32), 2)
    y2 = run(lambda: ops.conv3d(xd, pc, precision=1), 2)
    same, ssame = torch.equal(yo, yn) and torch.equal(yo, y2), torch.equal(so, sn)
    err = (yo - yn).abs().max().item()
    print(f"{'OK ' if same and ssame else 'BAD'} {N}x{Ci}->{Co} @{D}x{H}x{W}: bitwise equal {same} (max diff {err:.2e}) | stats equal {ssame}", flush=True)
    return same and ssame


def check_gnin(N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W) * 2 + 0.5
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    g, be = torch.rand(Ci) + 0.5, torch.randn(Ci) * 0.2
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    st = ops.groupnorm_stats(xd, 32)
    outs = {m: run(lambda: ops.conv3d_gn_in(xd, st, g.to(dev), be.to(dev), 32, pc), m) for m in (1, 2)}
    a, bb = outs[1], outs[2]
    a, bb = (a[0], bb[0]) if isinstance(a, tuple) else (a, bb)
    same = torch.equal(a, bb)
    print(f"{'OK ' if same else 'BAD'} gn-in {N}x{Ci}->{Co} @{D}x{H}x{W}: bitwise equal {same} (max diff {(a - bb).abs().max().item():.2e})", flush=True)
    return same


def timeit(N, Ci, Co, D, H, W, mode, iters=20):
    x = torch.randn(N, Ci, D, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
    def go():
        for _ in range(20): ops.conv3d(x, pc, precision=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.conv3d(x, pc, precision=1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    return run(go, mode)


if __name__ == "__main__":
    good = True
    if "--time-only" not in sys.argv:
        for case in [(2, 96, 96, 4, 8, 8), (1, 96, 96, 16, 64, 64), (2, 96, 192, 8, 32, 32), (1, 192, 96, 8, 32, 64), (8, 192, 192, 8, 32, 32),
                     (3, 16, 96, 4, 16, 8), (1, 256, 96, 8, 24, 40), (8, 384, 384, 4, 16, 16), (8, 192, 384, 4, 16, 16), (1, 96, 192, 8, 32, 32),
                     (5, 96, 96, 4, 8, 16)]:
            good &= check(*case)
        good &= check_gnin(2, 96, 96, 8, 32, 32)
        good &= check_gnin(1, 192, 192, 8, 16, 24)
        good &= check_gnin(3, 96, 96, 4, 8, 8)
        print("ALL OK" if good else "FAILURES", flush=True)
    for case in [(8, 96, 96, 16, 64, 64), (8, 192, 192, 8, 32, 32), (8, 384, 384, 4, 16, 16), (8, 96, 192, 8, 32, 32), (1, 96, 96, 16, 64, 64)]:
        r = []
        for rnd in range(3):
            r.append((timeit(*case, 1), timeit(*case, 2)))
        fl = 2.0 * case[0] * case[3] * case[4] * case[5] * case[1] * case[2] * 27 / 1e9
        print(f"{case}: role-split {min(a for a, _ in r):.4f} ms ({fl / min(a for a, _ in r):.0f} TF)  big-tile {min(b for _, b in r):.4f} ms ({fl / min(b for _, b in r):.0f} TF)   rounds {[(round(a, 4), round(b, 4)) for a, b in r]}", flush=True)
    sys.exit(0 if good else 1)
