"""time mphip_warp_sample_box (per-frame box of the source voxels a warp samples) at the graded size; checks it against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
B, D, H, W = 8, 16, 64, 64
coords = (torch.rand(B, D, H, W, 3, device=dev) * 4.5)
box = ops.warp_sample_box(coords)
f = coords.floor().to(torch.int32).reshape(B, -1, 3)
lo, hi = f.min(1).values, f.max(1).values
assert torch.equal(box[:, :3].cpu(), lo.cpu()), (box[:2], lo[:2])
for _ in range(5): ops.warp_sample_box(coords)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.warp_sample_box(coords)
e1.record(); torch.cuda.synchronize()
print(f"warp_sample_box B={B} {D}x{H}x{W}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (box of frame 0: {box[0].tolist()})")
