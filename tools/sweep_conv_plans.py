"""dev: per-layer sweep of the f16x3 planner's choices (tile variant x split-K) on G3d's level 1-3 convs at B=8.
Each configuration runs in its own process (the planner reads its env switches once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [  # name, Ci, Co, D, H, W
    ("L1 b1.conv1", 96, 192, 8, 32, 32), ("L1 b1.conv2", 192, 192, 8, 32, 32), ("L1 u4.conv1", 192, 96, 8, 32, 32), ("L1 u4.conv2", 96, 96, 8, 32, 32),
    ("L2 b2.conv1", 192, 384, 4, 16, 16), ("L2 b2.conv2", 384, 384, 4, 16, 16), ("L2 u2.conv1", 384, 192, 4, 16, 16), ("L2 u2.conv2", 192, 192, 4, 16, 16),
    ("L3 b3.conv1", 384, 768, 2, 8, 8), ("L3 b3.conv2", 768, 768, 2, 8, 8), ("L3 u0.conv1", 768, 384, 2, 8, 8), ("L3 u0.conv2", 384, 384, 2, 8, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from megaportrait_hack_amd import ops, _lib
    _lib.load()
    ci, co, d, h, w = (int(a) for a in sys.argv[2:7])
    dev = torch.device("cuda:0")
    x = torch.randn(8, ci, d, h, w, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02, torch.randn(co, device=dev))
    for _ in range(5): ops.conv3d(x, pc, precision=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.conv3d(x, pc, precision=1)
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 30 * 1e3:.1f}")
    sys.exit(0)
for name, ci, co, d, h, w in LAYERS:
    res = []
    for tile in ("", "0", "1", "2"):
        for sp in ("", "1", "2", "4", "8", "16"):
            if tile == "" and sp != "":
                continue
            if tile != "" and sp == "":
                continue
            env = dict(os.environ)
            if tile: env["MPHIP_F16X3_TILE"] = tile
            if sp: env["MPHIP_F16X3_SPLITS"] = sp
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(ci), str(co), str(d), str(h), str(w)], env=env, capture_output=True, text=True)
            out = r.stdout.strip().splitlines()
            us = float(out[-1]) if out and r.returncode == 0 else float("nan")
            res.append((us, f"tile={tile or 'auto'} sp={sp or 'auto'}"))
    base = res[0][0]
    best = min(res)
    gf = 2.0 * 8 * d * h * w * ci * co * 27 / 1e9
    print(f"{name:14s} {gf:6.1f} GF  auto {base:7.1f} us ({gf / base * 1e3:5.0f} TF/s)   best {best[0]:7.1f} us [{best[1]}]   " +
          " ".join(f"{u:.0f}" for u, _ in res[1:]), flush=True)
