"""Dev: run G3d forward+backward with the Winograd kernel on; every conv forward / bwd-data call is re-run on the SAME inputs with the
direct kernel and the relative difference printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib, model as M, autograd as AG
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = R.seeded_gbase_hot_state_dict(7)
x = R.seeded_tensor((B, 96, 16, 64, 64), 5, scale=1.3).to(dev)
dout = R.seeded_tensor((B, 96, 16, 64, 64), 6).to(dev)

def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)

def direct(fn, *a, **k):
    os.environ["MPHIP_WINOGRAD"] = "0"
    try:
        return fn(*a, **k)
    finally:
        os.environ["MPHIP_WINOGRAD"] = "1"

os.environ["MPHIP_WINOGRAD"] = "1"
orig_bwd, orig_fwd = ops.conv3d_bwd_data, ops.conv3d

def bwd_data(dy, pc_t, scale, *a, **k):
    got = orig_bwd(dy, pc_t, scale, *a, **k)
    want = direct(orig_bwd, dy, pc_t, scale, *a, **k)
    torch.cuda.synchronize()
    print(f"  bwd_data dy{tuple(dy.shape)} -> Co={pc_t.co}: winograd vs direct {rel(got, want):.1e}  |dy|max {dy.abs().max().item():.2e} scale {scale.flatten()[:2].tolist()}", flush=True)
    return got

def fwd(xx, pc, *a, **k):
    got = orig_fwd(xx, pc, *a, **k)
    want = direct(orig_fwd, xx, pc, *a, **k)
    g0, w0 = (got[0], want[0]) if isinstance(got, tuple) else (got, want)
    msg = f"  fwd x{tuple(xx.shape)} -> Co={pc.co} k={pc.k}: {rel(g0, w0):.1e}"
    if isinstance(got, tuple):
        msg += f" stats {rel(got[1], want[1]):.1e}"
    print(msg, flush=True)
    return got

ops.conv3d_bwd_data = bwd_data
ops.conv3d = fwd
g = M.G3d(96)
g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items() if k.startswith("G3d.")})
g = g.to(dev).train()
xi = x.clone().requires_grad_(True)
out = g(xi)
print("backward:")
out.backward(dout)
