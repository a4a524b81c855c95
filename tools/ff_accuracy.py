"""dev: the S2C warp field of the seeded fixture through the HIP generator vs the CPU oracle in fp32 and float64 (MPHIP_FF_FUSED=0/1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M, ops
dev = torch.device("cuda:0")
sd = R.seeded_gbase_hot_state_dict(7)
hot = M.GbaseHotSlice(); M.load_hot_state_dict(hot, sd); hot = hot.to(dev).eval()
inp = R.seeded_hot_inputs(1, 3)
args = [inp[k] for k in ("Rs", "ts", "zs", "es")]
with torch.no_grad():
    w1 = hot.warp_generator_s2c(*(a.to(dev) for a in args)).cpu().double()
    ref32 = R.warp_generator(*args, sd, "warp_generator_s2c.", True).double()
    ref64 = R.warp_generator(*(a.double() for a in args), {k: v.double() for k, v in sd.items()}, "warp_generator_s2c.", True)
print("fused" if ops._FF_FUSED else "old  ", "field: HIP vs fp32 oracle %.3e | vs float64: HIP %.3e, fp32 oracle %.3e | max %.3f" % (
    (w1 - ref32).abs().max(), (w1 - ref64).abs().max(), (ref32 - ref64).abs().max(), w1.abs().max()))
