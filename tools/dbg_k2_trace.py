"""dev: per-workgroup wall-clock stamps of K2's staged gather (library built with -DMPHIP_K2_TRACE).  usage: MPHIP_LIB=... python tools/dbg_k2_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from megaportrait_hack_amd import ops, _lib
from bench_warps import fields
lib = ctypes.CDLL(os.environ["MPHIP_LIB"])
B = 8
f = fields(B)["faithful"].cuda()
v = torch.randn(B, 96, 16, 64, 64, device="cuda")
for _ in range(5):
    ops.warp_volume(v, f)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4096 * 4))()
_lib.load().mphip_debug_k2_trace if False else None
h = ctypes.CDLL(os.environ["MPHIP_LIB"])
h.mphip_debug_k2_trace.argtypes = [ctypes.c_void_p]
assert _lib.load().__getattr__("mphip_debug_k2_trace")(buf) == 0 if hasattr(_lib.load(), "mphip_debug_k2_trace") else h.mphip_debug_k2_trace(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(4096, 4)[:256].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us
for name, col in (("start", 0), ("box known", 1), ("image staged", 2), ("done", 3)):
    c = t[:, col]
    print(f"{name:14s} min {c.min():6.1f}  median {np.median(c):6.1f}  p90 {np.percentile(c, 90):6.1f}  max {c.max():6.1f} us")
d = t[:, 1:] - t[:, :-1]
for name, col in (("prologue", 0), ("staging", 1), ("gather loop", 2)):
    c = d[:, col]
    print(f"{name:14s} min {c.min():6.1f}  median {np.median(c):6.1f}  p90 {np.percentile(c, 90):6.1f}  max {c.max():6.1f} us")
done = t[:, 3]
blk = np.arange(256)
print("done by XCD (blockIdx % 8):", " ".join(f"{done[blk % 8 == x].mean():5.1f}/{done[blk % 8 == x].max():5.1f}" for x in range(8)))
q = blk // 8   # order of arrival within an XCD
print("done by dispatch order within the XCD (octiles):", " ".join(f"{done[(q >= 16 * i) & (q < 16 * (i + 1))].mean():5.1f}" for i in range(8)))
loop = d[:, 2]
print("loop by dispatch order (octiles):", " ".join(f"{loop[(q >= 16 * i) & (q < 16 * (i + 1))].mean():5.1f}" for i in range(8)))
srt = np.sort(done)
print("done percentiles 10..100:", " ".join(f"{np.percentile(done, p):5.1f}" for p in range(10, 101, 10)))
