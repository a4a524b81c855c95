"""Dev check of the role-split F(2,3) conv against the lockstep F(2,3) kernel (bitwise expected on everything but rounding order: same
arithmetic), the fp64 truth, and timings of both.  usage: python tools/pp_check.py [--time-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"


def run(fn, pp):
    os.environ["MPHIP_WINO_PP"] = "1" if pp else "0"
    try:
        return fn()
    finally:
        os.environ.pop("MPHIP_WINO_PP", None)


def check(N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W) * 1.7
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    truth = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    yo, so = run(lambda: ops.conv3d(xd, pc, precision=1, gn_groups=32), False)
    yn, sn = run(lambda: ops.conv3d(xd, pc, precision=1, gn_groups=32), True)
    eo = (yo.cpu().double() - truth).abs().max().item()
    en = (yn.cpu().double() - truth).abs().max().item()
    same = torch.equal(yo, yn)
    ds = (so - sn).abs().max().item()
    ok = en < 2 * eo + 1e-6 and ds < 1e-5
    print(f"{'OK ' if ok else 'BAD'} {N}x{Ci}->{Co} @{D}x{H}x{W}: vs fp64 lockstep {eo:.2e} role-split {en:.2e} | bitwise equal {same} | stats diff {ds:.1e}", flush=True)
    return ok


def check_gnin(N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W) * 2 + 0.5
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    g, be = torch.rand(Ci) + 0.5, torch.randn(Ci) * 0.2
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    st = ops.groupnorm_stats(xd, 32)
    want = F.conv3d(F.relu(F.group_norm(x.double(), 32, g.double(), be.double(), 1e-5)), wt.double(), b.double(), padding=1)
    outs = {}
    for pp in (False, True):
        outs[pp] = run(lambda: ops.conv3d_gn_in(xd, st, g.to(dev), be.to(dev), 32, pc), pp)
    eo, en = ((outs[k].cpu().double() - want).abs().max().item() for k in (False, True))
    ok = en < 2 * eo + 1e-6
    print(f"{'OK ' if ok else 'BAD'} gn-in {N}x{Ci}->{Co} @{D}x{H}x{W}: lockstep {eo:.2e} role-split {en:.2e} | bitwise equal {torch.equal(outs[False], outs[True])}", flush=True)
    return ok


def timeit(N, Ci, Co, D, H, W, pp, iters=20):
    x = torch.randn(N, Ci, D, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
    def go():
        for _ in range(3): ops.conv3d(x, pc, precision=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.conv3d(x, pc, precision=1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    return run(go, pp)


if __name__ == "__main__":
    good = True
    if "--time-only" not in sys.argv:
        for case in [(2, 96, 96, 4, 8, 8), (1, 96, 96, 16, 64, 64), (2, 96, 192, 8, 32, 32), (1, 192, 96, 8, 32, 64), (8, 192, 192, 8, 32, 32),
                     (3, 16, 96, 4, 16, 8), (1, 256, 96, 8, 24, 40), (8, 384, 384, 4, 16, 16), (8, 192, 384, 4, 16, 16), (8, 384, 192, 4, 16, 16), (1, 96, 192, 8, 32, 32)]:
            good &= check(*case)
        good &= check_gnin(2, 96, 96, 8, 32, 32)
        good &= check_gnin(1, 192, 192, 8, 16, 24)
        good &= check_gnin(8, 384, 384, 4, 16, 16)
    for case in [(8, 96, 96, 16, 64, 64), (8, 96, 192, 8, 32, 32), (8, 192, 192, 8, 32, 32), (8, 192, 96, 8, 32, 32), (1, 96, 96, 16, 64, 64),
                 (8, 192, 384, 4, 16, 16), (8, 384, 384, 4, 16, 16), (8, 384, 192, 4, 16, 16), (1, 192, 192, 8, 32, 32)]:
        for rep in range(2):
            to, tn = timeit(*case, False), timeit(*case, True)
            fl = 2.0 * case[0] * case[3] * case[4] * case[5] * case[1] * case[2] * 27
            print(f"time {case}: lockstep {to:.3f} ms ({fl / to / 1e9:.0f} TF/s)  role-split {tn:.3f} ms ({fl / tn / 1e9:.0f} TF/s)  x{to / tn:.3f}", flush=True)
    print("ALL OK" if good else "FAILURES")
    sys.exit(0 if good else 1)
