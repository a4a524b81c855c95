"""Per-phase wall time of the role-split F(2,3) conv (conv3d_f16x3_wino_pp.hip; low-perturbation stamps after the barriers).  Needs a library
built with -DMPHIP_PP_PROFILE: tools/build_variant.sh ppprof conv3d_f16x3_wino_pp -DMPHIP_PP_PROFILE;
MPHIP_LIB=$PWD/build_variants/libmphip_ppprof.so python tools/prof_phases_pp.py [B Ci Co D H W]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPHIP_ALLOW_ABLATED", "1")   # dev tool: may be pointed at a timing variant (csrc/mphip_ablate.h)
import torch
from megaportrait_hack_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, Ci, Co, D, H, W = (int(a) for a in sys.argv[1:7]) if len(sys.argv) >= 7 else (8, 96, 96, 16, 64, 64)
os.environ["MPHIP_WINOGRAD_MIN_TILES"] = "1"
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
for _ in range(20): ops.conv3d(x, pc, precision=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.mphip_debug_wino_pp_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mphip_debug_wino_pp_profile(buf, 1)
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): ops.conv3d(x, pc, precision=1)
e1.record()
torch.cuda.synchronize()
lib.mphip_debug_wino_pp_profile(buf, 0)
v = list(buf)
tiles = B * (D // 4) * (H // 8) * (W // 8)
periods = N * (tiles / 256.0) * (Ci // 16)   # per workgroup (one per CU at full launches)
print(f"{B}x{Ci}->{Co} @{D}x{H}x{W}: {e0.elapsed_time(e1) / N:.3f} ms/launch (instrumented); cycles per phase, mean over waves and periods")
for team in (0, 1):
    t = v[team * 32: team * 32 + 21]; waves = max(t[20], 1); tot = sum(t[:18])
    per = [c / waves / periods * N for c in t[:18]]
    print(f" team {'AB'[team]}: cycles per wave and launch {tot / waves / N:.0f}  (MFMA(8) carries the epilogue of every tile = every {Ci // 16} periods)")
    print("   step      " + " ".join(f"{sp:6d}" for sp in range(9)))
    print("   LOAD wall " + " ".join(f"{per[2 * sp]:6.0f}" for sp in range(9)) + f"   sum {sum(per[0::2]):.0f}")
    print("   MFMA wall " + " ".join(f"{per[2 * sp + 1]:6.0f}" for sp in range(9)) + f"   sum {sum(per[1::2]):.0f}")
    ntile = N * tiles / 256.0
    ep = [c / waves / ntile * N for c in v[team * 32 + 21: team * 32 + 28]]
    print("   epilogue per tile: pre-issue+write0 %.0f | xform0 %.0f | write1 %.0f | xform1 %.0f | write2 %.0f | xform2 %.0f | to next LOAD %.0f   sum %.0f" % (*ep, sum(ep)))
