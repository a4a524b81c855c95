"""Soak: many back-to-back steps of the hot slice (inference and training) — throughput drift, allocator growth, finiteness.
usage: python tools/soak.py [infer_steps] [train_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M, training
from oracle import hotpath_ref as R

n_inf = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_tr = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
hot = M.GbaseHotSlice()
M.load_hot_state_dict(hot, R.seeded_gbase_hot_state_dict(7))
hot = hot.to(dev).eval()
inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(8, 21).items()}
with torch.no_grad():
    ref = hot(**inp).clone()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_reserved()
    for blk in range(6):
        t0 = time.perf_counter()
        for _ in range(n_inf // 6):
            out = hot(**inp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"infer block {blk}: {8 * (n_inf // 6) / dt:7.1f} frames/s, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB "
              f"(start {m0 / 2**20:.0f}), bitwise equal to the first step: {torch.equal(out, ref)}", flush=True)
hot.train()
opt = torch.optim.SGD(hot.parameters(), lr=1e-5)
tin = {k: v[:4].contiguous() for k, v in inp.items()}
loss_fn = lambda m, **kw: m(**kw).square().mean()
losses = []
for blk in range(3):
    t0 = time.perf_counter()
    for _ in range(n_tr // 3):
        losses.append(training.train_step(hot, loss_fn, opt, tin))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"train block {blk}: {(n_tr // 3) / dt:6.1f} steps/s, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB, "
          f"loss {float(losses[-n_tr // 3]):.6f} -> {float(losses[-1]):.6f}", flush=True)
assert all(torch.isfinite(l) for l in losses) and float(losses[-1]) < float(losses[0])
print("soak ok")
# two batches in flight on two streams (bench.py's default loop): every output equals the single-stream result bit for bit
hot.eval()
lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
with torch.no_grad():
    ref = hot(**inp).clone()
    for ln in lanes:
        ln.wait_stream(torch.cuda.current_stream())
    ok, t0 = True, time.perf_counter()
    outs = []
    for i in range(n_inf // 3):
        with torch.cuda.stream(lanes[i % 2]):
            outs.append(hot(**inp))
        if len(outs) == 64:
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(o, ref) for o in outs)
            outs = []
    torch.cuda.synchronize()
    ok = ok and all(torch.equal(o, ref) for o in outs)
    dt = time.perf_counter() - t0
    print(f"two streams, {n_inf // 3} steps: {8 * (n_inf // 3) / dt:7.1f} frames/s (incl. the checks), bitwise equal to the single-stream step: {ok}; "
          f"reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)
