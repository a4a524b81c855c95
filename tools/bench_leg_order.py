"""VERDICT r5 weak #2, second half: the steady-state autocast step is 67 ms with and without the conv policy (tools/e2e_autocast_bisect.py),
but bench.py's leg printed 112 ms.  Which earlier leg of the bench process makes the autocast leg slow, and on which steps?
usage: bench_leg_order.py <comma-separated prior legs: train,train_ac,reenact,e2e32,none>"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import megaportrait_hack_amd as pkg
from megaportrait_hack_amd import ops as _ops

pkg.request_hw_queues()
dev = torch.device("cuda:0")
prior = sys.argv[1].split(",") if len(sys.argv) > 1 else ["none"]
for p in prior:
    t0 = time.perf_counter()
    if p == "train":
        bench.train_leg(dev)
    elif p == "train_ac":
        bench.train_leg(dev, autocast=True)
    elif p == "reenact":
        bench.reenact_leg(dev, repeats=2, find=False)
    elif p == "e2e32":
        bench.end_to_end(dev, 8, steps=5, warmup=2)
    print(f"prior leg {p}: {time.perf_counter() - t0:.1f} s", flush=True)

from megaportrait_hack_amd import gbase
torch.manual_seed(20240501)
g = gbase.Gbase().to(dev).eval()
gen = torch.Generator(device="cpu").manual_seed(20240501)
xs = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
xd = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
per = []
with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g(xs, xd)
        torch.cuda.synchronize(); per.append(round((time.perf_counter() - t0) * 1e3, 1))
print(f"after {prior}: autocast e2e per-step ms {per}; cudnn.benchmark={torch.backends.cudnn.benchmark} "
      f"half_products_active={_ops.half_products_active()}", flush=True)
r = bench.end_to_end(dev, 8, steps=5, warmup=2, fp16=True)
print("bench.end_to_end(fp16) right after:", r["ms_per_step"], "ms/step", flush=True)
