"""dev: per-workgroup wall-clock stamps of the F(2,3) conv (library built with -DMPHIP_WN_TRACE): start and the end of every tile.
usage: MPHIP_LIB=build_variants/libmphip_wntrace.so python tools/dbg_wino_trace.py [B Ci Co D H W]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from megaportrait_hack_amd import ops, _lib
B, Ci, Co, D, H, W = (int(a) for a in sys.argv[1:7]) if len(sys.argv) >= 7 else (8, 96, 96, 16, 64, 64)
lib = _lib.load()
h = ctypes.CDLL(os.environ["MPHIP_LIB"])
h.mphip_debug_wino_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
xr = ops.tensor_range(x)
for _ in range(20):
    ops.conv3d(x, pc, precision=1, x_range=xr)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16))()
assert h.mphip_debug_wino_trace(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(1024, 16)[:256].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0
nt = int((t[0] > 0).sum())   # stamps written (start counts when > t0...)
print("start: min %.1f median %.1f max %.1f us" % (t[:, 0].min(), np.median(t[:, 0]), t[:, 0].max()))
ntile = 0
for i in range(1, 16):
    if (t[:, i] > 0).all():
        ntile = i
        print("tile %2d done: min %7.1f median %7.1f p90 %7.1f max %7.1f us   (tile time median %.1f  min %.1f  max %.1f)" % (
            i, t[:, i].min(), np.median(t[:, i]), np.percentile(t[:, i], 90), t[:, i].max(), np.median(t[:, i] - t[:, i - 1]), (t[:, i] - t[:, i - 1]).min(), (t[:, i] - t[:, i - 1]).max()))
end = t[:, ntile]
print("end by XCD (blockIdx %% 8), mean/max:", " ".join("%.1f/%.1f" % (end[np.arange(256) % 8 == k].mean(), end[np.arange(256) % 8 == k].max()) for k in range(8)))
print("end percentiles 0,10..100:", " ".join("%.1f" % np.percentile(end, p) for p in range(0, 101, 10)))
