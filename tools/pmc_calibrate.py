"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM):
  gn_stats      : pure 16 B/lane streaming read of a 402.7 MB tensor (> the 256 MB Infinity Cache)
  upsample 1x1x1: 4 B/lane streaming read + 4 B/lane streaming write of the same tensor
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
x = torch.randn(16, 96, 16, 64, 64, device=dev)
print("tensor bytes", x.numel() * 4)
for _ in range(2):
    ops.groupnorm_stats(x, 32)
    y = ops.upsample_nearest(x, (1, 1, 1))
torch.cuda.synchronize()
