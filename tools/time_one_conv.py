"""time one conv layer: [HALF=1] time_one_conv.py Ci Co D H W k B precision"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPHIP_ALLOW_ABLATED", "1")   # dev tool: may be pointed at a timing variant (csrc/mphip_ablate.h)
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
Ci, Co, D, H, W, k, B, prec = (int(a) for a in sys.argv[1:9])
if os.environ.get("HALF") == "1": _lib.load().mphip_conv3d_set_half_products(1)   # the autocast policy: one f16 product per multiply
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, k, k, k, device=dev) * 0.02, torch.randn(Co, device=dev))
for _ in range(40): ops.conv3d(x, pc, precision=prec)   # (warm: clocks and power state settle over ~20 launches)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): ops.conv3d(x, pc, precision=prec)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
print(f"{os.environ.get('MPHIP_LIB','default'):60s} {ms:.3f} ms  {2.0*B*D*H*W*Co*Ci*k**3/ms/1e9:.1f} TFLOP/s")
