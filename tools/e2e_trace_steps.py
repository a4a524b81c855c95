"""Runs N Gbase.forward steps under torch.autocast(float16) (for a kernel trace).  usage: e2e_trace_steps.py B steps policy(0/1)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megaportrait_hack_amd import gbase, ops
B, steps, policy = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if not policy:
    ops.autocast_half = lambda: False
dev = torch.device("cuda:0")
torch.manual_seed(1)
g = gbase.Gbase().to(dev).eval()
xs = torch.rand(B, 3, 512, 512, device=dev); xd = torch.rand(B, 3, 512, 512, device=dev)
with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
    for _ in range(steps):
        g(xs, xd)
torch.cuda.synchronize()
