import sys, os, collections, traceback
sys.path.insert(0, os.getcwd())
import torch
from megaportrait_hack_amd import model as M, ops
dev = torch.device("cuda:0")
hot = M.GbaseHotSlice().to(dev).eval()
B = 2
g = torch.Generator().manual_seed(1)
inp = dict(vs=torch.randn(B,96,16,64,64,generator=g), es=torch.randn(B,512,generator=g), zs=torch.randn(B,512,generator=g), zd=torch.randn(B,512,generator=g),
           Rs=torch.rand(B,3,generator=g)*60-30, Rd=torch.rand(B,3,generator=g)*60-30, ts=torch.randn(B,3,generator=g)*0.1, td=torch.randn(B,3,generator=g)*0.1)
inp = {k: v.to(dev) for k, v in inp.items()}
with torch.no_grad():
    hot(**inp); torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        hot(**inp); torch.cuda.synchronize()
evs = [e for e in prof.events() if 'copy' in e.name.lower() or 'Memcpy' in e.name or 'clone' in e.name or 'contiguous' in e.name]
c = collections.Counter(e.name for e in evs)
print(c)
for e in evs[:6]:
    print(e.name, e.stack[:6] if e.stack else None)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
