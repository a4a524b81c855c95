"""dev: how much of a step is the latency-bound head?  Times the B=8 hot slice (a) as is, (b) with the S2C field cached
(FlowField chain removed from the head of the step), (c) with both generator fields cached, (d) G3d alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M, ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(20240501)
hot = M.GbaseHotSlice().to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(20240501)
inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g), zs=torch.randn(B, 512, generator=g),
           zd=torch.randn(B, 512, generator=g), Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
           ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
inp = {k: v.to(dev) for k, v in inp.items()}


def timeit(fn, steps=40, warm=5):
    with torch.no_grad():
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("full step            %.3f ms" % timeit(lambda: hot(**inp)))
with torch.no_grad():
    ws = hot.warp_generator_s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])
    wc = hot.warp_generator_c2d(inp["Rd"], inp["td"], inp["zd"], inp["es"])
s2c_fwd, c2d_fwd = type(hot.warp_generator_s2c).forward, type(hot.warp_generator_c2d).forward
type(hot.warp_generator_s2c).forward = lambda self, *a: ws
print("S2C field cached     %.3f ms" % timeit(lambda: hot(**inp)))
type(hot.warp_generator_c2d).forward = lambda self, *a: wc
print("both fields cached   %.3f ms" % timeit(lambda: hot(**inp)))
with torch.no_grad():
    vc = ops.warp_volume(inp["vs"], ws)
print("G3d alone            %.3f ms" % timeit(lambda: hot.G3d(vc)))
print("K2 alone             %.3f ms" % timeit(lambda: ops.warp_volume(inp["vs"], ws)))
y = hot.G3d(vc) if False else None
with torch.no_grad():
    v2 = hot.G3d(vc)
print("K3 alone             %.3f ms" % timeit(lambda: ops.warp_volume_dsum(v2, wc)))
type(hot.warp_generator_s2c).forward, type(hot.warp_generator_c2d).forward = s2c_fwd, c2d_fwd
print("S2C generator alone  %.3f ms" % timeit(lambda: hot.warp_generator_s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])))
