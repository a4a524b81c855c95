"""VERDICT r5 weak #2, fourth step: catch the slow STATE of the autocast end-to-end step (tools/bench_leg_order.py: one Gbase instance ran
~103 ms/step for 10 steps, the next instance built in the same process 68 ms) without a profiler attached, and look at it from inside:
stage timings, clocks (rocm-smi), whether it survives a pause / an empty_cache / a rebuilt hot-slice plan.
usage: e2e_slow_state_hunt.py [rounds=5]"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import megaportrait_hack_amd as pkg
from megaportrait_hack_amd import gbase, model as M

pkg.request_hw_queues()
dev = torch.device("cuda:0")


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "memory)"))]
        return " | ".join(keep)[:600]
    except Exception as e:  # noqa: BLE001
        return repr(e)


def steps(g, xs, xd, n):
    per = []
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g(xs, xd)
            torch.cuda.synchronize(); per.append(round((time.perf_counter() - t0) * 1e3, 1))
    return per


def stages(g, xs, xd):
    parts = {}
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16):
        def timed(name, fn, n=3):
            fn(); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize(); parts[name] = round((time.perf_counter() - t) / n * 1e3, 2)
            return out
        ae = g.appearanceEncoder
        trunk = timed("trunk2d", lambda: ae.trunk2d(xs))
        vol = trunk.view(8, 96, 16, *trunk.shape[2:])
        def tail(v):
            for name in M.Eapp3DTail._ORDER:
                v = getattr(ae, name)(v)
            return v
        vs = timed("tail3d", lambda: tail(vol))
        es = timed("descriptor", lambda: ae.descriptor(xs))
        Rs, ts, zs = timed("Emtn", lambda: g.motionEncoder(xs))
        hot = timed("hot", lambda: g.hot_slice(vs, es, Rs, ts, zs, Rs, ts, zs))
        img = timed("G2d", lambda: g.G2d(hot))
        timed("pyramid", lambda: g.image_pyramid(img))
    return parts


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for r in range(rounds):
    bench.train_leg(dev, autocast=True)
    bench.end_to_end(dev, 8, steps=5, warmup=2)
    t_build = time.perf_counter()
    torch.manual_seed(20240501)
    g = gbase.Gbase().to(dev).eval()
    gen = torch.Generator(device="cpu").manual_seed(20240501)
    xs = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
    xd = torch.rand(8, 3, 512, 512, generator=gen).to(dev)
    t_build = time.perf_counter() - t_build
    per = steps(g, xs, xd, 6)
    slow = sorted(per[1:])[len(per[1:]) // 2] > 85
    print(f"round {r}: build {t_build:.1f} s; per-step ms {per}; {'SLOW' if slow else 'normal'}; reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
    if slow:
        print("  smi (right after):", smi(), flush=True)
        print("  stages:", stages(g, xs, xd), flush=True)
        print("  6 more steps:", steps(g, xs, xd, 6), flush=True)
        # is it the hot-slice plan's side stream (fork / join across HIP streams)?  one-stream plan, then the per-op schedule, then back
        g.overlap_generators = False
        print("  one-stream plan (overlap_generators=False):", steps(g, xs, xd, 5), flush=True)
        g.overlap_generators = True
        print("  two-stream plan again (the cached one):", steps(g, xs, xd, 4), flush=True)
        g.use_c_plan = False
        print("  per-op Python schedule (use_c_plan=False):", steps(g, xs, xd, 5), flush=True)
        g.use_c_plan = True
        g.__dict__.pop("_plans", None)
        print("  a NEW two-stream plan for the same instance:", steps(g, xs, xd, 5), flush=True)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            print("  the same instance driven from another caller stream:", steps(g, xs, xd, 5), flush=True)
        time.sleep(3)
        print("  after a 3 s pause:", steps(g, xs, xd, 4), flush=True)
        g.__dict__.pop("_plans", None)   # the hot-slice plan (its streams, packs, workspace) is rebuilt by the next forward
        torch.cuda.empty_cache()
        print("  after empty_cache:", steps(g, xs, xd, 4), "reserved", torch.cuda.memory_reserved() >> 20, flush=True)
        xs2, xd2 = xs.clone(), xd.clone()
        print("  with cloned inputs:", steps(g, xs2, xd2, 4), flush=True)
        g2 = gbase.Gbase().to(dev).eval()
        print("  a second instance beside it:", steps(g2, xs, xd, 5), " and the first again:", steps(g, xs, xd, 4), flush=True)
        print("  stages of the second:", stages(g2, xs, xd), flush=True)
        del g2
    else:
        print("  stages:", stages(g, xs, xd), flush=True)
    del g
    torch.cuda.empty_cache()
