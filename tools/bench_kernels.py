"""Per-kernel timing on the GPU (HIP events on torch's current stream): G3d conv layers, GN, warps.
Dev tool — not the graded bench (bench.py)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib

dev = torch.device("cuda:0")
_lib.load()

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layers = [  # Ci, Co, D, H, W, k, count in G3d
    (96, 96, 16, 64, 64, 3, 3), (96, 192, 8, 32, 32, 3, 1), (192, 192, 8, 32, 32, 3, 1), (96, 192, 8, 32, 32, 1, 1),
    (192, 384, 4, 16, 16, 3, 1), (384, 384, 4, 16, 16, 3, 2), (192, 384, 4, 16, 16, 1, 1),
    (384, 768, 2, 8, 8, 3, 1), (768, 768, 2, 8, 8, 3, 1), (384, 768, 2, 8, 8, 1, 1),
    (768, 384, 2, 8, 8, 3, 1), (768, 384, 2, 8, 8, 1, 1),
    (384, 192, 4, 16, 16, 3, 1), (192, 192, 4, 16, 16, 3, 1), (384, 192, 4, 16, 16, 1, 1),
    (192, 96, 8, 32, 32, 3, 1), (96, 96, 8, 32, 32, 3, 1), (192, 96, 8, 32, 32, 1, 1),
]
tot_ms = 0; tot_fl = 0
for Ci, Co, D, H, W, k, cnt in layers:
    x = torch.randn(B, Ci, D, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(Co, Ci, k, k, k, device=dev) * 0.02, torch.randn(Co, device=dev))
    ms = timeit(lambda: ops.conv3d(x, pc))
    fl = 2.0 * B * D * H * W * Co * Ci * k ** 3
    tot_ms += ms * cnt; tot_fl += fl * cnt
    print(f"conv k{k} {Ci:4d}->{Co:4d} @{D}x{H}x{W} B={B}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  (x{cnt})")
print(f"G3d convs total: {tot_ms:.2f} ms per batch of {B} -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s, {B / tot_ms * 1e3:.1f} frames/s (convs only)")

x = torch.randn(B, 96, 16, 64, 64, device=dev)
g = torch.ones(96, device=dev); b = torch.zeros(96, device=dev)
ms = timeit(lambda: ops.groupnorm_stats(x, 32)); print(f"gn_stats full-res: {ms:.3f} ms  {x.numel()*4/ms/1e6:.0f} GB/s")
st = ops.groupnorm_stats(x, 32)
ms = timeit(lambda: ops.groupnorm_apply(x, st, g, b, 32, relu=True)); print(f"gn_apply full-res: {ms:.3f} ms  {x.numel()*8/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.groupnorm_apply(x, st, g, b, 32, residual=x, relu=True, pool2=True)); print(f"gn_apply+res+pool: {ms:.3f} ms  {x.numel()*(8+0.5)/ms/1e6:.0f} GB/s")
xh = torch.randn(B, 96, 8, 32, 32, device=dev)
ms = timeit(lambda: ops.upsample_trilinear2(xh)); print(f"upsample2 ->full-res: {ms:.3f} ms  {x.numel()*4*1.125/ms/1e6:.0f} GB/s")
field = torch.randn(B, 3, 64, 64, 64, device=dev) * 1.3
ms = timeit(lambda: ops.warp_volume(x, field)); print(f"warp_volume (faithful field): {ms:.3f} ms  {(x.numel()*8+field.numel()*4)/ms/1e6:.0f} GB/s algorithmic")
ms = timeit(lambda: ops.warp_volume_dsum(x, field)); print(f"warp_volume_dsum (faithful): {ms:.3f} ms  {(x.numel()*4*(1+1/16)+field.numel()*4)/ms/1e6:.0f} GB/s algorithmic")
wide = torch.rand(B, 3, 64, 64, 64, device=dev) * torch.tensor([66., 66., 18.], device=dev).view(1, 3, 1, 1, 1) - 1.5
ms = timeit(lambda: ops.warp_volume(x, wide)); print(f"warp_volume (stress field): {ms:.3f} ms  {(x.numel()*8+field.numel()*4)/ms/1e6:.0f} GB/s algorithmic")
ms = timeit(lambda: ops.warp_volume_dsum(x, wide)); print(f"warp_volume_dsum (stress): {ms:.3f} ms  {(x.numel()*4*(1+1/16)+field.numel()*4)/ms/1e6:.0f} GB/s algorithmic")
th = torch.randn(B, 3, 4, device=dev); em = torch.rand(B, 3, 16, 16, 16, device=dev)
ms = timeit(lambda: ops.warp_field_compose(th, em)); print(f"warp_field_compose: {ms:.3f} ms  {field.numel()*4/ms/1e6:.0f} GB/s")
