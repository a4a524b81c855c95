"""Times conv3d_bwd_weight (both precisions) on G3d's layer shapes.  usage: time_bwd_weight.py [B] [--l0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
layers = [(96, 96, 16, 64, 64, 3), (192, 192, 8, 32, 32, 3), (384, 384, 4, 16, 16, 3), (768, 768, 2, 8, 8, 3), (96, 192, 8, 32, 32, 1)]
if "--l0" in sys.argv: layers = layers[:1]
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for Ci, Co, D, H, W, k in layers:
    x = torch.randn(B, Ci, D, H, W, device=dev); dy = torch.randn(B, Co, D, H, W, device=dev) * 1e-3
    _, sc = ops.grad_prep(dy)
    fl = 2.0 * B * D * H * W * Co * Ci * k ** 3
    for prec in (1, 0):
        ms = timeit(lambda: ops.conv3d_bwd_weight(x, dy, k, sc, precision=prec))
        print(f"bwd_weight k{k} {Ci}->{Co} @{D}x{H}x{W} B={B} prec={prec}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: ops.grad_prep(dy)); print(f"   grad_prep: {ms:.3f} ms  {dy.numel() * 4 / ms / 1e6:.0f} GB/s")
