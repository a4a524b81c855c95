"""Repro loop for the stale-loss read of training.GraphedTrainStep (VERDICT r4 weak #1 / ADVICE r4): N fresh (eager, graphed) pairs walk the
sequence of tests/test_gpu_backward.py::test_graphed_train_step_matches_eager WITHOUT a device-wide synchronize before the loss is read;
a replay whose loss differs from the eager step's is classified (equal to the previous replay's value = a stale read; anything else = a
wrong step).  usage: python tools/repro_graph_loss.py [N=20] [mode]   mode: raw (read static_loss directly) | call (what __call__ returns)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import model as M, training, _lib
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "call"


def make():
    g = M.G3d(96)
    sd = R.seeded_state_dict(R.g3d_shapes(96), 81, prefix="G3d.")
    g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items()})
    return g.to(dev).train()


x = R.seeded_tensor((1, 96, 8, 16, 16), 82).to(dev)
tgt = R.seeded_tensor((1, 96, 8, 16, 16), 83).to(dev)
loss_fn = lambda m, x: F.mse_loss(m(x), tgt)
stale = wrong = total = 0
for it in range(N):
    eager, graphed = make(), make()
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-2, momentum=0.9)
    opt_g = torch.optim.SGD(graphed.parameters(), lr=1e-2, momentum=0.9)
    step = training.GraphedTrainStep(graphed, loss_fn, opt_g, {"x": x}, warmup=2)
    prev = None
    for k in range(7):
        if k < 3:
            le = training.train_step(eager, loss_fn, opt_e, {"x": x}); lg = step(x=x)
        else:
            lg = step(x=x); le = training.train_step(eager, loss_fn, opt_e, {"x": x})
        g = (step.static_loss if mode == "raw" else lg).item()   # (no torch.cuda.synchronize() in front)
        e = le.item()
        total += 1
        if abs(e - g) > 1e-5 * abs(e):
            if prev is not None and abs(g - prev) <= 1e-7 * abs(prev):
                stale += 1
            else:
                wrong += 1
            print(f"  run {it} step {k}: eager {e:.8f} graph {g:.8f} previous replay {prev}", flush=True)
        prev = g
print(f"{mode}: {total} replays read, {stale} stale reads, {wrong} wrong steps")
sys.exit(0 if stale + wrong == 0 else 1)
