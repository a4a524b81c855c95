"""dev: per-step time of the hot slice through the C-side plan vs the per-op Python schedule (eager), B = 1, 2, 8."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M

dev = torch.device("cuda:0")
torch.manual_seed(20240501)
hot = M.GbaseHotSlice().to(dev).eval()
for B in (1, 2, 8):
    g = torch.Generator(device="cpu").manual_seed(20240501)
    inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g), zs=torch.randn(B, 512, generator=g),
               zd=torch.randn(B, 512, generator=g), Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
               ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    res = {}
    for name, flag in (("python", False), ("c-plan", True)):
        hot.use_c_plan = flag
        with torch.no_grad():
            for _ in range(5):
                out = hot(**inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                out = hot(**inp)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 40 * 1e3
            # latency of ONE isolated step (GPU idle before it)
            lat = []
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = hot(**inp)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0) * 1e3)
            res[name + "_latency"] = sorted(lat)[len(lat) // 2]
            res[name + "_out"] = out.clone()
    same = torch.equal(res["python_out"], res["c-plan_out"])
    print(f"B={B}: python {res['python']:.3f} ms/step (isolated {res['python_latency']:.3f}), c-plan {res['c-plan']:.3f} ms/step "
          f"(isolated {res['c-plan_latency']:.3f}), {B / res['c-plan'] * 1e3:.0f} frames/s, bitwise equal: {same}")
