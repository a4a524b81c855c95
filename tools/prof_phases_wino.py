"""Phase timing of the Winograd f16x3 conv (s_memtime instrumentation).  Needs a library built with -DMPHIP_PROFILE_PHASES:
MPHIP_EXTRA_FLAGS=-DMPHIP_PROFILE_PHASES MPHIP_BUILD_DIR=/tmp/b bash megaportrait-hack_amd/csrc/build.sh build_variants/libmphip_prof.so;
MPHIP_LIB=$PWD/build_variants/libmphip_prof.so python tools/prof_phases_wino.py [B Ci Co D H W]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPHIP_ALLOW_ABLATED", "1")   # dev tool: may be pointed at a timing variant (csrc/mphip_ablate.h)
import torch
from megaportrait_hack_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, Ci, Co, D, H, W = (int(a) for a in sys.argv[1:7]) if len(sys.argv) >= 7 else (8, 96, 96, 16, 64, 64)
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
for _ in range(3): ops.conv3d(x, pc, precision=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.mphip_debug_f16x3_wino_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mphip_debug_f16x3_wino_profile(buf, 1)
N = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): ops.conv3d(x, pc, precision=1)
e1.record()
torch.cuda.synchronize()
lib.mphip_debug_f16x3_wino_profile(buf, 0)
v = list(buf); waves = v[7]; tot = sum(v[:7])
names = ["prologue", "interval: DMA issue + frag reads + MFMA issue", "interval: wait own DMA pieces", "interval: barrier", "halo write (transform+split+ds_write)",
         "halo barrier + reload", "output transform + epilogue"]
print(f"{B}x{Ci}->{Co} @{D}x{H}x{W}: {e0.elapsed_time(e1) / N:.3f} ms/launch (instrumented); waves {waves}, cycles per wave {tot / waves:.0f}")
for n, c in zip(names, v[:7]): print(f"  {n:48s} {c / waves:10.0f} cycles/wave  {100 * c / tot:5.1f} %")
