"""Side measurement: where the end-to-end Gbase.forward time goes and what the cheap PyTorch-ROCm knobs buy (MIOpen find mode,
channels_last).  Usage: python tools/e2e_probe.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, fp16, bench, cl, steps=6, warmup=3):
    from megaportrait_hack_amd import gbase

    torch.backends.cudnn.benchmark = bench
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    g = gbase.Gbase().to(dev).eval()
    if cl:
        for m in (g.appearanceEncoder, g.motionEncoder, g.G2d):
            for p in m.parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
    xs = torch.rand(B, 3, 512, 512, device=dev)
    xd = torch.rand(B, 3, 512, 512, device=dev)
    if cl:
        xs, xd = xs.contiguous(memory_format=torch.channels_last), xd.contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16, enabled=fp16):
        for _ in range(warmup):
            g(xs, xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g(xs, xd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # parts
        parts = {}
        def timed(name, fn):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize(); parts[name] = round((time.perf_counter() - t) / 3 * 1e3, 2)
            return out
        vs, es = timed("Eapp", lambda: g.appearanceEncoder(xs))
        Rs, ts, zs = timed("Emtn(xs)", lambda: g.motionEncoder(xs))
        hot = timed("hot", lambda: g.hot_slice(vs, es, Rs, ts, zs, Rs, ts, zs))
        timed("G2d", lambda: g.G2d(hot))
    print(f"B={B} fp16={fp16} miopen_find={bench} channels_last={cl}: {dt*1e3:.1f} ms/step  {B/dt:.1f} frames/s  parts(ms)={parts}", flush=True)
    del g
    torch.cuda.empty_cache()


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for fp16 in (False, True):
        for bench, cl in ((False, False), (True, False), (True, True)):
            try:
                run(B, fp16, bench, cl)
            except Exception as e:  # noqa: BLE001
                print("failed", fp16, bench, cl, repr(e)[:300], flush=True)
