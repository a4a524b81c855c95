"""Runs one conv layer a few times (for rocprofv3 --pmc passes). usage: run_one_conv.py Ci Co D H W k B iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
Ci, Co, D, H, W, k, B, iters = (int(a) for a in sys.argv[1:9])
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, k, k, k, device=dev) * 0.02, torch.randn(Co, device=dev))
for _ in range(iters):
    y = ops.conv3d(x, pc)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
