"""Dev check of the two-frame (depth-2) mode of the big-tile F(2,3) conv against the fp64 truth and the direct f16x3 kernel
(MPHIP_WINOGRAD_D2=0), then interleaved timings at G3d's 2x8x8 level.  usage: python tools/d2_check.py [--time-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"


def run(fn, d2):
    os.environ["MPHIP_WINOGRAD_D2"] = "1" if d2 else "0"
    try:
        return fn()
    finally:
        os.environ.pop("MPHIP_WINOGRAD_D2", None)


def check(N, Ci, Co, H, W, fused=False):
    x = torch.randn(N, Ci, 2, H, W) * 1.7 + (0.5 if fused else 0.0)
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    assert run(lambda: lib.mphip_conv3d_kernel_variant(N, Ci, Co, 2, H, W, 3, 1), True) == 5, "two-frame mode not taken"
    assert run(lambda: lib.mphip_conv3d_kernel_variant(N, Ci, Co, 2, H, W, 3, 1), False) != 5
    if fused:
        g, be = torch.rand(Ci) + 0.5, torch.randn(Ci) * 0.2
        st = ops.groupnorm_stats(xd, 32)
        truth = F.conv3d(F.relu(F.group_norm(x.double(), 32, g.double(), be.double(), 1e-5)), wt.double(), b.double(), padding=1)
        f = lambda: ops.conv3d_gn_in(xd, st, g.to(dev), be.to(dev), 32, pc)
    else:
        truth = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
        f = lambda: ops.conv3d(xd, pc, precision=1, gn_groups=32)
    yo, yn = run(f, False), run(f, True)
    so = sn = None
    if isinstance(yo, tuple): (yo, so), (yn, sn) = yo, yn
    top = truth.abs().max().item()
    eo, en = (yo.cpu().double() - truth).abs().max().item() / top, (yn.cpu().double() - truth).abs().max().item() / top
    ds = 0.0 if so is None else (so - sn).abs().max().item()
    ok = en < 2 * eo + 1e-6 and en < 2e-5 and ds < 1e-4
    print(f"{'OK ' if ok else 'BAD'} {'gn-in ' if fused else ''}{N}x{Ci}->{Co} @2x{H}x{W}: vs fp64 direct {eo:.2e} two-frame F(2,3) {en:.2e} | stats diff {ds:.1e}", flush=True)
    return ok


def timeit(N, Ci, Co, H, W, d2, iters=30):
    x = torch.randn(N, Ci, 2, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
    def go():
        for _ in range(20): ops.conv3d(x, pc, precision=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.conv3d(x, pc, precision=1)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    return run(go, d2)


if __name__ == "__main__":
    good = True
    if "--time-only" not in sys.argv:
        for case in [(2, 96, 96, 8, 8), (8, 384, 768, 8, 8), (8, 768, 768, 8, 8), (8, 768, 384, 8, 8), (8, 384, 384, 8, 8), (3, 96, 96, 8, 16), (1, 96, 192, 16, 8),
                     (5, 192, 96, 8, 8), (64, 96, 96, 8, 8)]:
            good &= check(*case)
        for case in [(2, 96, 96, 8, 8), (8, 768, 768, 8, 8), (8, 384, 384, 8, 8), (3, 192, 96, 8, 16), (1, 96, 96, 8, 8)]:
            good &= check(*case, fused=True)
        print("ALL OK" if good else "FAILURES", flush=True)
    os.environ.pop("MPHIP_WINOGRAD_MIN_TILES", None)
    tot = [0.0, 0.0]
    for case in [(8, 384, 768, 8, 8), (8, 768, 768, 8, 8), (8, 768, 384, 8, 8), (8, 384, 384, 8, 8), (1, 768, 768, 8, 8)]:
        r = [(timeit(*case, False), timeit(*case, True)) for _ in range(3)]
        a, b = min(v for v, _ in r), min(v for _, v in r)
        if case[0] == 8: tot[0] += a; tot[1] += b
        print(f"{case}: direct {a:.1f} us  two-frame F(2,3) {b:.1f} us   (incl. split-K reduce; variant {run(lambda: lib.mphip_conv3d_kernel_variant(case[0], case[1], case[2], 2, case[3], case[4], 3, 1), True)})", flush=True)
    print(f"the level's four convs at B=8: direct {tot[0]:.1f} us, two-frame {tot[1]:.1f} us", flush=True)
    sys.exit(0 if good else 1)
