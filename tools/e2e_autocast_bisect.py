"""where do the +42 ms per B=8 step of `end_to_end_autocast_fp16` (114 -> 71 frames/s) come from?
Times Gbase.forward under torch.autocast(float16) step by step (one-off costs show as outlier steps) and stage by stage, with the
conv arithmetic policy of model._autocast_policy on and off (ops.autocast_half patched to False = r04's behaviour).
Usage: python tools/e2e_autocast_bisect.py [B] [policy|nopolicy|both]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, policy, steps=8):
    from megaportrait_hack_amd import gbase, ops

    real = ops.autocast_half
    if not policy:
        ops.autocast_half = lambda: False
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    g = gbase.Gbase().to(dev).eval()
    xs = torch.rand(B, 3, 512, 512, device=dev)
    xd = torch.rand(B, 3, 512, 512, device=dev)
    per_step = []
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
        for _ in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g(xs, xd)
            torch.cuda.synchronize()
            per_step.append(round((time.perf_counter() - t0) * 1e3, 1))
        parts = {}

        def timed(name, fn, n=3):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            parts[name] = round((time.perf_counter() - t) / n * 1e3, 2)
            return out

        ae = g.appearanceEncoder
        trunk = timed("Eapp.trunk2d", lambda: ae.trunk2d(xs))
        vol = trunk.view(B, 96, 16, *trunk.shape[2:])
        from megaportrait_hack_amd import model as M

        def tail(v):
            for name in M.Eapp3DTail._ORDER:
                v = getattr(ae, name)(v)
            return v

        vs = timed("Eapp.tail3d", lambda: tail(vol))
        es = timed("Eapp.descriptor", lambda: ae.descriptor(xs))
        Rs, ts, zs = timed("Emtn(xs)", lambda: g.motionEncoder(xs))
        Rd, td, zd = g.motionEncoder(xd)
        hot = timed("hot", lambda: g.hot_slice(vs, es, Rs, ts, zs, Rd, td, zd))
        img = timed("G2d", lambda: g.G2d(hot))
        timed("pyramid", lambda: g.image_pyramid(img))
        timed("forward", lambda: g(xs, xd))
    ops.autocast_half = real
    print(f"B={B} policy={policy}: per-step ms {per_step}  parts(ms)={parts}  dtypes: trunk {trunk.dtype} vs {vs.dtype} "
          f"es {es.dtype} Rs {Rs.dtype} hot {hot.dtype} img {img.dtype}", flush=True)
    del g
    torch.cuda.empty_cache()


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    for policy in ((True, False) if which == "both" else ((which == "policy"),)):
        run(B, policy)
