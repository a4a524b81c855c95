"""Throughput of the hot slice at BASELINE config 1's frame size (256x256 -> volume 96x16x32x32), B frames per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hot = M.GbaseHotSlice().to(dev).eval()
inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(B, 3, D=16, H=32, W=32).items()}
with torch.no_grad():
    for _ in range(5): hot.forward_any_size(**inp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = hot.forward_any_size(**inp)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"256x256 frames (volume 96x16x32x32) B={B}: {ms:.3f} ms/step  {B / ms * 1e3:.1f} frames/s")
