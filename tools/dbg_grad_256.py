"""Dev: test_hot_slice_backward's vs-gradient: HIP (F(2,3) on / off) and the fp32 CPU oracle against the fp64 truth, and the truth's own
movement under several perturbation samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import _lib, model as M
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")
sd = R.seeded_gbase_hot_state_dict(7)
inp = R.seeded_hot_inputs(1, 43, D=16, H=32, W=32)

def truth(noise, seed, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    t_in = {k: v.clone().to(dtype) for k, v in inp.items()}
    if noise:
        t_in["vs"] = t_in["vs"] * (1 + noise * torch.randn(t_in["vs"].shape, generator=gen, dtype=dtype))
    t_in = {k: v.requires_grad_(True) for k, v in t_in.items()}
    t_sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    o = R.hot_slice(sd=t_sd, **t_in)
    o.backward(R.seeded_tensor(tuple(o.shape), 92).to(dtype))
    return t_in, t_sd

def rel(a, b):
    d = a.detach().cpu().double() - b.detach().double()
    return d.abs().max().item() / b.detach().double().abs().max().item(), d.norm().item() / b.detach().double().norm().item()

t_in, t_sd = truth(0, 0)
print("fp32 CPU oracle vs fp64: vs %s" % (rel(truth(0, 0, torch.float32)[0]["vs"].grad, t_in["vs"].grad),))
for noise, seed in ((1e-6, 1), (2e-6, 2), (2e-6, 3), (5e-6, 4), (5e-6, 5), (1e-5, 6)):
    p_in, p_sd = truth(noise, seed)
    worst = max((rel(p_sd[n].grad, t_sd[n].grad)[0], n) for n in t_sd if t_sd[n].grad is not None and n.startswith("G3d"))
    print(f"truth moved by noise {noise:g} (seed {seed}): vs max/L2 {rel(p_in['vs'].grad, t_in['vs'].grad)}  worst G3d param {worst}")
for wino in ("1", "0"):
    os.environ["MPHIP_WINOGRAD"] = wino
    hot = M.GbaseHotSlice(); M.load_hot_state_dict(hot, sd); hot = hot.to(dev).train()
    g_in = {k: v.to(dev).requires_grad_(True) for k, v in inp.items()}
    out = hot.forward_any_size(**g_in)
    out.backward(R.seeded_tensor(tuple(out.shape), 92).to(dev))
    worst = max((rel(p.grad, t_sd[n].grad)[0], n) for n, p in hot.named_parameters() if t_sd[n].grad is not None and n.startswith("G3d"))
    print(f"HIP winograd={wino} vs fp64: vs max/L2 {rel(g_in['vs'].grad, t_in['vs'].grad)}  worst G3d param {worst}")
