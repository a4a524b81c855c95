"""Config-3-sized gradient check against an fp64 truth: HIP (f16x3 / fp32) and the fp32 CPU oracle are both compared with
CPU autograd of the oracle run in float64.  usage: python tools/grad_truth.py [B] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:
    ops.set_conv_precision(sys.argv[2])
dev = torch.device("cuda:0")
sd = R.seeded_gbase_hot_state_dict(7)
inp = R.seeded_hot_inputs(B, 47)


def cpu(dtype):
    i = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in inp.items()}
    s = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    t0 = time.time()
    out = R.hot_slice(sd=s, **i)
    dout = R.seeded_tensor(tuple(out.shape), 93).to(dtype)
    out.backward(dout)
    print(f"cpu {dtype}: {time.time() - t0:.1f} s", flush=True)
    g = {k: v.grad for k, v in s.items() if v.grad is not None}
    g.update({"in." + k: v.grad for k, v in i.items()})
    return out.detach(), g


o64, g64 = cpu(torch.float64)
o32, g32 = cpu(torch.float32)
hot = M.GbaseHotSlice()
M.load_hot_state_dict(hot, sd)
hot = hot.to(dev).train()
gi = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in inp.items()}
out = hot(**gi)
out.backward(R.seeded_tensor(tuple(out.shape), 93).to(dev))
gh = {n: p.grad for n, p in hot.named_parameters() if p.grad is not None}
gh.update({"in." + k: v.grad for k, v in gi.items()})
print("forward max-abs vs fp64: hip %.3e  cpu32 %.3e" % ((out.detach().cpu().double() - o64).abs().max().item(),
                                                          (o32.double() - o64).abs().max().item()))
rows = []
for k, t in g64.items():
    sc = t.abs().max().item() or 1e-30
    eh = (gh[k].detach().cpu().double() - t).abs().max().item() / sc
    ec = (g32[k].double() - t).abs().max().item() / sc
    rows.append((eh, ec, k, sc))
rows.sort(reverse=True)
print("worst 15 by HIP error (rel. to max-abs of the fp64 gradient):  hip | cpu-fp32 | name | max-abs")
for eh, ec, k, sc in rows[:15]:
    print(f"  {eh:.3e}  {ec:.3e}  {k}  {sc:.3e}")
print("worst 5 by cpu-fp32 error:")
for eh, ec, k, sc in sorted(rows, key=lambda r: -r[1])[:5]:
    print(f"  {eh:.3e}  {ec:.3e}  {k}  {sc:.3e}")
