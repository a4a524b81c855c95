"""Where do the ~12 hipMemcpyDtoD blits (__amd_rocclr_copyBuffer) per step come from?  Logs every Tensor.copy_/clone/contiguous/
to/expand-materialising call on CUDA tensors during one hot-slice step, with the first caller frame inside the package."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M, _lib
_lib.load()
dev = torch.device("cuda:0")
hot = M.GbaseHotSlice().to(dev).eval()
g = torch.Generator().manual_seed(0)
B = 8
inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g), zs=torch.randn(B, 512, generator=g),
           zd=torch.randn(B, 512, generator=g), Rs=torch.rand(B, 3, generator=g) * 60 - 30, Rd=torch.rand(B, 3, generator=g) * 60 - 30,
           ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
inp = {k: v.to(dev) for k, v in inp.items()}
with torch.no_grad():
    hot(**inp); hot(**inp)
torch.cuda.synchronize()
hits = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if isinstance(self, torch.Tensor) and self.is_cuda:
            copied = name in ("copy_", "clone") or (name == "contiguous" and out.data_ptr() != self.data_ptr()) or \
                     (name in ("to", "float") and isinstance(out, torch.Tensor) and out.data_ptr() != self.data_ptr())
            if copied:
                fr = [x for x in traceback.extract_stack()[:-1] if "megaportrait" in x.filename]
                hits[(name, tuple(self.shape), f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?")] += 1
        return out
    setattr(torch.Tensor, name, f)
for n in ("copy_", "clone", "contiguous", "to", "float"):
    wrap(n)
with torch.no_grad():
    hot(**inp)
torch.cuda.synchronize()
for k, v in hits.most_common():
    print(v, k)
print("total python-level copies:", sum(hits.values()))
