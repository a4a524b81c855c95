"""Runs K2 and K3 a few times on the BASELINE shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
B = 16  # 403 MB volume: larger than the 256 MB Infinity Cache
x = torch.randn(B, 96, 16, 64, 64, device=dev)
field = torch.randn(B, 3, 64, 64, 64, device=dev) * 1.3
for _ in range(3):
    y = ops.warp_volume(x, field)
    z = ops.warp_volume_dsum(x, field)
torch.cuda.synchronize()
print("ok")
