"""dev: the k=1 shortcut convs of G3d at batch B, per in-workgroup channel split (MPHIP_F16X3_K1_KS): time_k1.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from megaportrait_hack_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
LAYERS = [("down1 96->192", 96, 192, 8, 32, 32), ("down2 192->384", 192, 384, 4, 16, 16), ("down3 384->768", 384, 768, 2, 8, 8),
          ("up0 768->384", 768, 384, 2, 8, 8), ("up1 384->192", 384, 192, 4, 16, 16), ("up2 192->96", 192, 96, 8, 32, 32)]


def time_one(x, pc, prec):
    for _ in range(3): y = ops.conv3d(x, pc, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): y = ops.conv3d(x, pc, precision=prec)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3, y


for B in [int(a) for a in sys.argv[1:]] or [8, 1]:
    print(f"--- B={B}")
    for name, ci, co, d, h, w in LAYERS:
        x = torch.randn(B, ci, d, h, w, device=dev)
        wt = torch.randn(co, ci, 1, 1, 1, device=dev) * 0.05
        bias = torch.randn(co, device=dev)
        pc = ops.PackedConv(wt, bias)
        ref = torch.einsum("oc,ncdhw->nodhw", wt[:, :, 0, 0, 0].double(), x.double()) + bias.double().view(1, -1, 1, 1, 1)
        line = f"{name:16s}"
        t32, _ = time_one(x, pc, 0)
        line += f" fp32 {t32:6.1f}"
        if not _lib.load().mphip_conv3d_supported(B, ci, co, d, h, w, 1, 1):
            print(line + "  (f16x3 k=1 not supported at this size)"); continue
        for ks in ("1", "4", "8", ""):
            if ks: os.environ["MPHIP_F16X3_K1_KS"] = ks
            else: os.environ.pop("MPHIP_F16X3_K1_KS", None)
            try:
                t, y = time_one(x, pc, 1)
                err = float((y.double() - ref).abs().max())
                line += f" | ks={ks or 'auto'} {t:6.1f} err {err:.1e}"
            except Exception as e:
                line += f" | ks={ks or 'auto'} n/a"
        os.environ.pop("MPHIP_F16X3_K1_KS", None)
        print(line, flush=True)
