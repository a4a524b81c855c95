"""dev: the stale graphed loss, counted.  G3d training step as a hipGraph vs the same steps eager: parameters always
agree; the LOSS the graph returns is stale when the graph holds a MEMSET node (ATen's multi-block reduction) that is not rewritten.
usage: [MPHIP_GRAPH_MEMSET_FIX=0|1] [MPHIP_BATCHED_PACKS=0|1] dbg_replay_stale_loss.py mse|staged [stash] [redo]"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from oracle import hotpath_ref as R
from megaportrait_hack_amd import model as M, training
dev = torch.device("cuda:0")
sd = R.seeded_state_dict(R.g3d_shapes(96), 91, prefix="G3d.")
x = R.seeded_tensor((1, 96, 8, 16, 16), 92).to(dev)
def mk():
    g = M.G3d(96); g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items()}); return g.to(dev).train()
tgt = R.seeded_tensor((1, 96, 8, 16, 16), 93).to(dev)
kind = sys.argv[1]
stash = {}
def loss_fn(m, x):
    y = m(x)
    if "stash" in sys.argv: stash["y"] = y
    if kind == "mse":
        return F.mse_loss(y, tgt)
    d = (y - tgt).square().view(-1, 1024)          # staged: no multi-block (semaphore) reduction
    return d.sum(1).sum() / y.numel()
gt, ge = mk(), mk()
opt_g = torch.optim.SGD(gt.parameters(), lr=1e-3); opt_e = torch.optim.SGD(ge.parameters(), lr=1e-3)
step = training.GraphedTrainStep(gt, loss_fn, opt_g, {"x": x}, warmup=2)
ystat = stash.get("y")
bad = 0
for i in range(12):
    scale = (100.0, 1.0, 1.0, 100.0)[i % 4]
    lg = step(x=x * scale)
    redo = F.mse_loss(ystat.detach(), tgt).item() if (ystat is not None and "redo" in sys.argv) else -1
    le = training.train_step(ge, loss_fn, opt_e, {"x": x * scale})
    ok = abs(lg.item() - le.item()) <= 1e-5 * abs(le.item())
    bad += not ok
    print(kind, os.environ.get("MPHIP_BATCHED_PACKS"), i, "graph loss", lg.item(), "eager", le.item(), "recomputed from the graph's y", redo, "OK" if ok else "STALE")
print("SUMMARY", sys.argv[1:], os.environ.get("MPHIP_BATCHED_PACKS"), "stale", bad, "of 12")
