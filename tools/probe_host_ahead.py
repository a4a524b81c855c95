"""dev: is the host ahead of the GPU in the plan's steady state?  Host time per forward call without synchronisation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M
B = 8
dev = torch.device("cuda:0")
torch.manual_seed(20240501)
hot = M.GbaseHotSlice().to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(20240501)
inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g), zs=torch.randn(B, 512, generator=g),
           zd=torch.randn(B, 512, generator=g), Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
           ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
inp = {k: v.to(dev) for k, v in inp.items()}
with torch.no_grad():
    for _ in range(5):
        hot(**inp)
    torch.cuda.synchronize()
    ts = []
    t_all = time.perf_counter()
    for _ in range(20):
        t0 = time.perf_counter()
        out = hot(**inp)
        ts.append((time.perf_counter() - t0) * 1e3)
    t_issue = (time.perf_counter() - t_all) * 1e3
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t_all) * 1e3
print("host ms per call:", " ".join(f"{t:.2f}" for t in ts))
print(f"issue of 20 steps took {t_issue:.1f} ms, GPU finished after {t_total:.1f} ms")
