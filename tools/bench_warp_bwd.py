"""Dev timing of mphip_warp_volume_bwd at the training shard's size (B=4, 96x16x64x64) for fields whose samples stay in a box of
E^3 source voxels (E = 3, 4, 5: the dense MFMA dv path) and for a field that travels (the tiled scatter).
usage: python tools/bench_warp_bwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megaportrait_hack_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.manual_seed(3)
v = torch.randn(B, 96, 16, 64, 64, device=dev)
u = torch.rand(B, 3, 64, 64, 64, device=dev)
fields = {
    "E=3 (coords in [0,2))": u * 2.4 - 0.4,
    "E=4 (coords in [0,3))": u * 3.4 - 0.4,
    "E=5 (coords in [0,4))": u * 4.4 - 0.4,
    "travelling (tiled scatter)": u * 2.0 + torch.stack(torch.meshgrid(torch.linspace(0, 12, 64), torch.linspace(0, 50, 64),
                                                                    torch.linspace(0, 50, 64), indexing="ij")).flip(0)[None].to(dev),
}
for dsum in (False, True):
    dout = torch.randn((B, 96, 64, 64) if dsum else (B, 96, 16, 64, 64), device=dev)
    for name, f in fields.items():
        for want_field in (False, True):
            for _ in range(3):
                ops.warp_volume_bwd(v, f, dout, dsum, True, want_field)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.warp_volume_bwd(v, f, dout, dsum, True, want_field)
            e1.record()
            torch.cuda.synchronize()
            print(f"dsum={int(dsum)} {name:28s} dfield={int(want_field)}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
