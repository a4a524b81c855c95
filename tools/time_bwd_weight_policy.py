"""time the f16x3 3x3x3 bwd-weight launch (+ its slab reduce) under both arithmetic policies: time_bwd_weight_policy.py Ci Co D H W B"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
Ci, Co, D, H, W, B = (int(a) for a in sys.argv[1:7])
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, D, H, W, device=dev)
dy = torch.randn(B, Co, D, H, W, device=dev)
_, scale = ops.grad_prep(dy, want_bias=False)
for half in (False, True, False, True):
    with ops.half_products(half):
        for _ in range(10): ops.conv3d_bwd_weight(x, dy, 3, scale, precision=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d_bwd_weight(x, dy, 3, scale, precision=1)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{Ci}->{Co} @{D}x{H}x{W} B={B} half_products={int(half)}: {ms:.3f} ms  {2.0*B*D*H*W*Co*Ci*27/ms/1e9:.1f} TFLOP/s")
