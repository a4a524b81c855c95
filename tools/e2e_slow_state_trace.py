"""VERDICT r5 weak #2, third step.  tools/bench_leg_order.py found a STATE, not a code path: after some leg sequences one Gbase instance runs
every autocast step at ~103 ms while the next instance built in the same process runs 68 ms.  This script reproduces that inside ONE
process under `rocprofv3 --kernel-trace` and brackets the two instances' steps with marker kernels (erfinv: used nowhere else), so that
tools/e2e_slow_state_split.py can compare per-kernel durations slow instance vs normal instance.
usage: e2e_slow_state_trace.py <prior legs>"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import megaportrait_hack_amd as pkg
from megaportrait_hack_amd import gbase

pkg.request_hw_queues()
dev = torch.device("cuda:0")
for p in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    {"train": lambda: bench.train_leg(dev), "train_ac": lambda: bench.train_leg(dev, autocast=True),
     "reenact": lambda: bench.reenact_leg(dev, repeats=2, find=False), "e2e32": lambda: bench.end_to_end(dev, 8, steps=5, warmup=2),
     "none": lambda: None}[p]()
mark = torch.rand(4096, device=dev) * 0.5


def instance(tag):
    torch.manual_seed(20240501)
    g = gbase.Gbase().to(dev).eval()
    gen = torch.Generator(device="cpu").manual_seed(20240501)
    xs = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
    xd = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
    per = []
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
        for i in range(6):
            if i == 2:
                torch.erfinv(mark)          # marker: steps 2-5 of this instance follow
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g(xs, xd)
            torch.cuda.synchronize(); per.append(round((time.perf_counter() - t0) * 1e3, 1))
    torch.erfinv(mark)                      # marker: end
    torch.cuda.synchronize()
    ptrs = sorted(p.data_ptr() for p in g.parameters())
    print(f"{tag}: per-step ms {per}; parameter addresses {ptrs[0]:#x}..{ptrs[-1]:#x}; reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
    del g
    torch.cuda.empty_cache()


instance("instance 1")
instance("instance 2")
