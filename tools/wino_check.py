"""Dev check of the Winograd F(2,3) f16x3 conv (conv3d_f16x3_wino.hip) against the direct f16x3 kernel, the exact-fp32 kernel and the
fp64 truth, plus timings of both kernels on G3d's level-0/1 layers.  usage: python tools/wino_check.py [--time-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import ops, _lib

_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def conv(x, pc, wino, **kw):
    os.environ["MPHIP_WINOGRAD"] = "1" if wino else "0"
    try:
        return ops.conv3d(x, pc, precision=1, **kw)
    finally:
        os.environ.pop("MPHIP_WINOGRAD", None)


def check(N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W) * 1.7
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    b = torch.randn(Co) * 0.1
    truth = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    pc = ops.PackedConv(wt.to(dev), b.to(dev))
    xd = x.to(dev)
    e32 = (ops.conv3d(xd, pc, precision=0).cpu().double() - truth).abs().max().item()
    ed = (conv(xd, pc, False).cpu().double() - truth).abs().max().item()
    yw, st = conv(xd, pc, True, gn_groups=32)
    ew = (yw.cpu().double() - truth).abs().max().item()
    tr = truth.reshape(N, 32, -1)
    em = (st[:, 0].cpu().double() - tr.mean(-1).reshape(-1)).abs().max().item()
    er = (st[:, 1].cpu().double() - (1.0 / torch.sqrt(tr.var(-1, unbiased=False) + 1e-5)).reshape(-1)).abs().max().item()
    ok = ew < 3 * e32 + 1e-6 and em < 1e-5 and er < 1e-4
    print(f"{'OK ' if ok else 'BAD'} {N}x{Ci}->{Co} @{D}x{H}x{W}: vs fp64 fp32-MFMA {e32:.2e} direct {ed:.2e} winograd {ew:.2e} | stats mean {em:.1e} rstd {er:.1e}", flush=True)
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
    for wino in (False, True):
        os.environ["MPHIP_WINOGRAD"] = "1" if wino else "0"
        outs[wino] = (ops.conv3d_gn_in(xd, st, g.to(dev), be.to(dev), 32, pc).cpu().double() - want).abs().max().item()
    os.environ.pop("MPHIP_WINOGRAD", None)
    ok = outs[True] < 3 * outs[False] + 1e-6
    print(f"{'OK ' if ok else 'BAD'} gn-in {N}x{Ci}->{Co} @{D}x{H}x{W}: direct {outs[False]:.2e} winograd {outs[True]:.2e}", flush=True)
    return ok


def timeit(N, Ci, Co, D, H, W, wino, iters=20):
    os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"   # (timing: force the F(2,3) kernel wherever its tiling applies)
    x = torch.randn(N, Ci, D, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
    for _ in range(3):
        conv(x, pc, wino)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    os.environ["MPHIP_WINOGRAD"] = "1" if wino else "0"
    e0.record()
    for _ in range(iters):
        ops.conv3d(x, pc, precision=1)
    e1.record()
    torch.cuda.synchronize()
    os.environ.pop("MPHIP_WINOGRAD", None)
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    good = True
    if "--time-only" not in sys.argv:
        for case in [(2, 96, 96, 4, 8, 8), (1, 96, 96, 16, 64, 64), (2, 96, 192, 8, 32, 32), (1, 192, 96, 8, 32, 64), (8, 192, 192, 8, 32, 32),
                     (3, 16, 96, 4, 16, 8), (1, 256, 96, 8, 24, 40), (8, 384, 384, 4, 16, 16), (8, 192, 384, 4, 16, 16), (8, 384, 192, 4, 16, 16), (1, 96, 192, 8, 32, 32)]:
            os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"
            good &= check(*case)
        good &= check_gnin(2, 96, 96, 8, 32, 32)
        good &= check_gnin(1, 192, 192, 8, 16, 24)
        good &= check_gnin(8, 384, 384, 4, 16, 16)
        os.environ.pop("MPHIP_WINOGRAD_MIN_TILES", None)
    for case in [(8, 96, 96, 16, 64, 64), (8, 96, 192, 8, 32, 32), (8, 192, 192, 8, 32, 32), (8, 192, 96, 8, 32, 32), (1, 96, 96, 16, 64, 64),
                 (4, 96, 96, 16, 64, 64), (8, 192, 384, 4, 16, 16), (8, 384, 384, 4, 16, 16), (8, 384, 192, 4, 16, 16), (8, 192, 192, 4, 16, 16),
                 (1, 96, 192, 8, 32, 32), (1, 192, 192, 8, 32, 32), (1, 192, 96, 8, 32, 32), (2, 96, 192, 8, 32, 32), (2, 384, 384, 4, 16, 16)]:
        for rep in range(2):
            td, tw = timeit(*case, False), timeit(*case, True)
            fl = 2.0 * case[0] * case[3] * case[4] * case[5] * case[1] * case[2] * 27
            print(f"time {case}: direct {td:.3f} ms ({fl / td / 1e9:.0f} TF/s)  winograd {tw:.3f} ms ({fl / tw / 1e9:.0f} TF/s)  x{td / tw:.3f}", flush=True)
    print("ALL OK" if good else "FAILURES")
    sys.exit(0 if good else 1)
