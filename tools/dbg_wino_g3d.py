"""Dev: G3d under autograd with the Winograd kernel on vs off — first module whose output / grad_output differs beyond rounding."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib, model as M
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = R.seeded_gbase_hot_state_dict(7)
x = R.seeded_tensor((B, 96, 16, 64, 64), 5, scale=1.3).to(dev)
dout = R.seeded_tensor((B, 96, 16, 64, 64), 6).to(dev)

def run(wino, eps=0.0):
    os.environ["MPHIP_WINOGRAD"] = str(wino)
    g = M.G3d(96)
    g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items() if k.startswith("G3d.")})
    g = g.to(dev).train()
    fwd, bwd = {}, {}
    for name, mod in g.named_modules():
        if name == "":
            continue
        mod.register_forward_hook(lambda m, i, o, name=name: fwd.__setitem__(name, o.detach().clone()) if torch.is_tensor(o) else None)
        mod.register_full_backward_hook(lambda m, gi, go, name=name: bwd.__setitem__(name, go[0].detach().clone()) if go and torch.is_tensor(go[0]) else None)
    xi = (x * (1 + eps * torch.randn_like(x))).requires_grad_(True) if eps else x.clone().requires_grad_(True)
    out = g(xi)
    out.backward(dout)
    grads = {n: p.grad.detach().clone() for n, p in g.named_parameters()}
    return fwd, bwd, grads, xi.grad.detach().clone()

def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)

f0, b0, g0, dx0 = run(0)
f1, b1, g1, dx1 = run(1)
f0b, b0b, g0b, _ = run(0, 1e-6)   # direct kernels, input perturbed by 1e-6 relative noise
print("module outputs (forward order), rel diff winograd vs direct [direct vs direct with 1e-6 input noise]:")
for n in f0:
    if n in f1:
        print(f"  fwd {n:32s} {rel(f1[n], f0[n]):.1e}  [{rel(f0b[n], f0[n]):.1e}]")
print("grad_outputs (backward order):")
for n in b0:
    if n in b1:
        print(f"  bwd {n:32s} {rel(b1[n], b0[n]):.1e}  [{rel(b0b[n], b0[n]):.1e}]")
print("parameter grads:")
for n in g0:
    print(f"  {n:36s} {rel(g1[n], g0[n]):.1e}  [{rel(g0b[n], g0[n]):.1e}]")
print("dx", rel(dx1, dx0))
