"""Phase timing of the dominant f16x3 conv (s_memtime instrumentation).  Needs a library built with
-DMPHIP_PROFILE_PHASES:  MPHIP_EXTRA_FLAGS=-DMPHIP_PROFILE_PHASES MPHIP_BUILD_DIR=/tmp/b bash megaportrait-hack_amd/csrc/build.sh /path/lib.so;
MPHIP_LIB=/path/lib.so python tools/prof_phases.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPHIP_ALLOW_ABLATED", "1")   # dev tool: may be pointed at a timing variant (csrc/mphip_ablate.h)
import torch
from megaportrait_hack_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B = 8
x = torch.randn(B, 96, 16, 64, 64, device=dev)
pc = ops.PackedConv(torch.randn(96, 96, 3, 3, 3, device=dev) * 0.02, torch.randn(96, device=dev))
for _ in range(3): ops.conv3d(x, pc, precision=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.mphip_debug_f16x3_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mphip_debug_f16x3_profile(buf, 1)
for _ in range(5): ops.conv3d(x, pc, precision=1)
torch.cuda.synchronize()
lib.mphip_debug_f16x3_profile(buf, 0)
v = list(buf); waves = v[7]; tot = sum(v[:7])
names = ["prologue", "X-load issue", "W-DMA issue", "tap loop (LDS reads + MFMA)", "group barrier wait", "X write + barrier", "epilogue"]
print(f"waves {waves}, cycles per wave {tot / waves:.0f}")
for n, c in zip(names, v[:7]): print(f"  {n:32s} {c / waves:10.0f} cycles/wave  {100 * c / tot:5.1f} %")
