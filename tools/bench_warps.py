"""K2 (warp_volume) / K3 (warp_volume_dsum) in isolation on the BASELINE volume (96x16x64x64) for four kinds of warp field:
  faithful  : what the reference's generators produce (samples land in the 4^3 low corner)
  smooth    : pixel-space identity + a smooth displacement of +-3 voxels in x/y, +-1 slice in z (a warp that uses the whole volume)
  rot30     : rigid 30 deg rotation about the volume centre + the same displacement (a large head turn in voxel units)
  noise     : every voxel samples an independent uniform position of the volume (tests/_fields()["wide"]: no coherence)
usage: python tools/bench_warps.py [B] [iters] [--json out.json] [--only kind]   (run under rocprofv3 --pmc for HBM counters)
Algorithmic bytes: K2 = read v + field + write out = 53.5 MB/frame, K3 = 29.9 MB/frame."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib

C, D, H, W = 96, 16, 64, 64
K2_BYTES = (C * D * H * W * 2 + 3 * 64 ** 3) * 4     # per frame
K3_BYTES = (C * D * H * W + 3 * 64 ** 3 + C * H * W) * 4


def fields(B):
    g = torch.Generator(device="cpu").manual_seed(7)
    lin = lambda n: torch.linspace(-1, 1, n)
    # the field is given on a 64^3 grid and resized to (D,H,W) with align_corners=True: build it at (64,64,64)
    zz, yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(64.0), torch.arange(64.0), indexing="ij")
    # smooth displacement: 3^3 control points (correlation length ~32 voxels), +-3 voxels in x/y, +-1 slice in z
    amp = torch.tensor([3.0, 3.0, 1.0]).view(1, 3, 1, 1, 1)
    smooth = torch.nn.functional.interpolate((torch.rand(B, 3, 3, 3, 3, generator=g) * 2 - 1) * amp, size=(64, 64, 64), mode="trilinear",
                                             align_corners=True)
    gx, gy, gz = lin(64)[None, None, :].expand(64, 64, 64), lin(64)[None, :, None].expand(64, 64, 64), lin(64)[:, None, None].expand(64, 64, 64)
    # sample coordinate (pixel units) = identity-grid value + field (model.py:1052-1062): field = target - g
    tgt_id = torch.stack((xx, yy, zz * (15.0 / 63.0)))                      # voxel's own position in the 16x64x64 volume
    ident = (tgt_id - torch.stack((gx, gy, gz)))[None] + smooth
    a = math.radians(30.0)
    cx, cy = 31.5, 31.5
    rx = cx + (xx - cx) * math.cos(a) - (yy - cy) * math.sin(a)
    ry = cy + (xx - cx) * math.sin(a) + (yy - cy) * math.cos(a)
    rot = (torch.stack((rx, ry, zz * (15.0 / 63.0))) - torch.stack((gx, gy, gz)))[None] + smooth
    noise = (torch.rand(B, 3, 64, 64, 64, generator=g)) * torch.tensor([66.0, 66.0, 18.0]).view(1, 3, 1, 1, 1) - 2.0
    faithful = torch.randn(B, 3, 64, 64, 64, generator=g) * 0.75 + 0.4
    return {"faithful": faithful, "smooth": ident.contiguous(), "rot30": rot.contiguous(), "noise": noise}


def measure(B, iters=20, kinds=None, quiet=False, field_override=None):
    """-> {"K2 warp_volume / kind": {...}, "K3 warp_volume_dsum / kind": {...}}: HIP-event timing on the current stream."""
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn(B, C, D, H, W, device=dev)
    res = {}
    table = fields(B) if field_override is None else field_override
    for kind, f in table.items():
        if kinds and kind not in kinds:
            continue
        f = f.to(dev)
        for name, fn, nbytes in (("K2 warp_volume", ops.warp_volume, K2_BYTES), ("K3 warp_volume_dsum", ops.warp_volume_dsum, K3_BYTES)):
            for _ in range(3):
                fn(x, f)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(x, f)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            gbps = B * nbytes / ms / 1e6
            res[f"{name} / {kind}"] = {"ms": round(ms, 4), "algorithmic_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4), "B": B}
            if not quiet:
                print(f"{name:22s} {kind:9s} B={B}: {ms * 1e3:8.1f} us  {gbps / 1e3:6.2f} TB/s algorithmic = {gbps / 80:5.1f} % of 8 TB/s")
    return res


if __name__ == "__main__":
    _lib.load()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 8
    iters = int(args[1]) if len(args) > 1 else 20
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    res = measure(B, iters, kinds=[only] if only else None)
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
