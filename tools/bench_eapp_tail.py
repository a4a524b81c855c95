"""Eapp 3D tail (row f1, model.py:271-290: six ResBlock3D_Adaptive(96,96) applications on 96x16x64x64) through the HIP
path vs the same graph in PyTorch-ROCm eager.  usage: bench_eapp_tail.py [B] [--torch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
sd = R.seeded_state_dict(R.eapp_tail_shapes(), 17, prefix="appearanceEncoder.")
feat = torch.randn(B, 1536, 64, 64, device=dev)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tail = M.Eapp3DTail()
tail.load_state_dict({k[len("appearanceEncoder."):]: v for k, v in sd.items()})
tail = tail.to(dev).eval()
with torch.no_grad():
    ms = timed(lambda: tail(feat))
print(f"HIP   Eapp 3D tail B={B}: {ms:.2f} ms  {B / ms * 1e3:.1f} frames/s  ({391.0 * B / ms:.0f} TFLOP/s algorithmic)")
if "--torch" in sys.argv:
    gsd = {k: v.to(dev) for k, v in sd.items()}
    with torch.no_grad():
        ms = timed(lambda: R.eapp_tail3d(feat, gsd), iters=3)
    print(f"torch Eapp 3D tail B={B}: {ms:.2f} ms  {B / ms * 1e3:.1f} frames/s")
