"""dev: N steps of the hot slice through the C-side plan (for rocprofv3 traces): run_plan_steps.py [B] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
torch.manual_seed(20240501)
hot = M.GbaseHotSlice().to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(20240501)
inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g), zs=torch.randn(B, 512, generator=g),
           zd=torch.randn(B, 512, generator=g), Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
           ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
inp = {k: v.to(dev) for k, v in inp.items()}
with torch.no_grad():
    for _ in range(steps):
        out = hot(**inp)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
