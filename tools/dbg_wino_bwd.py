"""Dev: bwd-data through the Winograd kernel (transposed packs) vs the direct kernel vs fp64 truth; then config 3's per-tensor gradient
errors with the transformed-domain kernel on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import ops, _lib, model as M
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(1)

def bwd(N, Ci, Co, D, H, W, mag=1e-3, like=False):
    # conv: Ci -> Co forward; bwd-data maps dy [N,Co,...] -> dx [N,Ci,...]
    wt = torch.randn(Co, Ci, 3, 3, 3) * (Ci * 27) ** -0.5
    dy = torch.randn(N, Co, D, H, W) * mag
    dy[:, :, : D // 2] *= 1e-3     # wide dynamic range across the volume, like a real gradient
    truth = F.conv_transpose3d(dy.double(), wt.double(), padding=1)
    wd = wt.to(dev)
    if like:
        pc_f = ops.PackedConv(wd, None)
        pc_f.packed(1)
        pc_t = ops.PackedConv(wd, None, transposed=True, header_from=pc_f)
    else:
        pc_t = ops.PackedConv(wd, None, transposed=True)
    dyd = dy.to(dev)
    _, scale = ops.grad_prep(dyd, want_bias=False)
    res = {}
    for wino in (0, 1):
        os.environ["MPHIP_WINOGRAD"] = str(wino)
        dx = ops.conv3d_bwd_data(dyd, pc_t, scale)
        res[wino] = (dx.cpu().double() - truth).abs().max().item() / truth.abs().max().item()
    os.environ.pop("MPHIP_WINOGRAD")
    print(f"bwd-data like={like} {N}x{Co}->{Ci} @{D}x{H}x{W}: rel err direct {res[0]:.2e} winograd {res[1]:.2e}", flush=True)


def rel(a, b):
    return (a.detach().cpu().double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-30)

def rel2(a, b):
    return (a.detach().cpu().double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-300)

sd = R.seeded_gbase_hot_state_dict(7)
inp = R.seeded_hot_inputs(4, 47)
cpu_in = {k: v.double().requires_grad_(True) for k, v in inp.items()}
cpu_sd = {k: v.double().requires_grad_(True) for k, v in sd.items()}
out_ref = R.hot_slice(sd=cpu_sd, **cpu_in)
dout = R.seeded_tensor(tuple(out_ref.shape), 93)
out_ref.backward(dout.double())
# the truth's own sensitivity: the same fp64 graph with vs perturbed by 1e-6 relative noise
torch.manual_seed(3)
p_in = {k: (v.double() * (1 + 1e-6 * torch.randn_like(v.double()))).requires_grad_(True) if k == "vs" else v.double().requires_grad_(True) for k, v in inp.items()}
p_sd = {k: v.double().requires_grad_(True) for k, v in sd.items()}
R.hot_slice(sd=p_sd, **p_in).backward(dout.double())
g3d = [n for n in sd if n.startswith("G3d.") and cpu_sd[n].grad is not None]
print("fp64 truth vs fp64 truth with vs*(1+1e-6 noise): G3d parameter gradients  max-abs rel: max %.1e median %.1e | L2 rel: max %.1e" % (
    max(rel(p_sd[n].grad, cpu_sd[n].grad) for n in g3d), sorted(rel(p_sd[n].grad, cpu_sd[n].grad) for n in g3d)[len(g3d) // 2],
    max(rel2(p_sd[n].grad, cpu_sd[n].grad) for n in g3d)))
print("   worst:", ", ".join(f"{n}={rel(p_sd[n].grad, cpu_sd[n].grad):.1e}" for n in sorted(g3d, key=lambda n: -rel(p_sd[n].grad, cpu_sd[n].grad))[:5]))
for wino in (0, 1):
    os.environ["MPHIP_WINOGRAD"] = str(wino)
    hot = M.GbaseHotSlice()
    M.load_hot_state_dict(hot, sd)
    hot = hot.to(dev).train()
    gpu_in = {k: v.clone().to(dev).requires_grad_(True) for k, v in inp.items()}
    out = hot(**gpu_in)
    print(f"winograd={wino}: forward max-abs err {(out.detach().cpu().double() - out_ref.detach()).abs().max().item():.2e}")
    out.backward(dout.to(dev))
    errs = sorted(((rel(p.grad, cpu_sd[n].grad), n) for n, p in hot.named_parameters() if cpu_sd[n].grad is not None and p.grad is not None), reverse=True)
    print("  worst parameter gradients:", ", ".join(f"{n}={e:.1e}" for e, n in errs[:8]))
    e2 = sorted(((rel2(p.grad, cpu_sd[n].grad), n) for n, p in hot.named_parameters() if n.startswith("G3d.") and cpu_sd[n].grad is not None), reverse=True)
    em = sorted(((rel(p.grad, cpu_sd[n].grad), n) for n, p in hot.named_parameters() if n.startswith("G3d.") and cpu_sd[n].grad is not None), reverse=True)
    print("  G3d: max-abs rel worst", ", ".join(f"{n}={e:.1e}" for e, n in em[:4]), "| median %.1e" % em[len(em) // 2][0])
    print("  G3d: L2 rel worst", ", ".join(f"{n}={e:.1e}" for e, n in e2[:4]), "| median %.1e" % e2[len(e2) // 2][0])
    print("  inputs:", ", ".join(f"{k}={rel(gpu_in[k].grad, cpu_in[k].grad):.1e}" for k in inp))
