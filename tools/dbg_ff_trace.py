"""dev: per-workgroup wall-clock stamps of FlowField's ROW block kernel (library built with -DMPHIP_FF_TRACE).
usage: MPHIP_LIB=build_variants/libmphip_fftrace.so python tools/dbg_ff_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from megaportrait_hack_amd import model as M, _lib
lib = _lib.load()
h = ctypes.CDLL(os.environ["MPHIP_LIB"])
h.mphip_debug_ff_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
ff = M.FlowField().to(dev).eval()
z = torch.randn(8, 512, device=dev)
with torch.no_grad():
    for _ in range(5):
        ff(z)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 8))()
assert h.mphip_debug_ff_trace(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(512, 8)[:256, :6].astype(np.float64)
t = (t - t[:, 0].min()) / 100.0
names = ["start", "conv loop done", "fold done", "residual loop done", "(fold) before finish", "end"]
for i, nme in enumerate(names):
    print("%-22s min %6.1f median %6.1f max %6.1f us" % (nme, t[:, i].min(), np.median(t[:, i]), t[:, i].max()))
d = t[:, 1:] - t[:, :-1]
print("phases (median): conv loop %.1f  fold %.1f  residual loop %.1f  fold %.1f  finish %.1f us" % tuple(np.median(d[:, i]) for i in range(5)))
