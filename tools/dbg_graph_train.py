"""Dev: the loss sequence of tests/test_gpu_backward.py::test_graphed_train_step_matches_eager, eager vs graph, step by step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from megaportrait_hack_amd import model as M, training, _lib
from oracle import hotpath_ref as R
_lib.load()
dev = torch.device("cuda:0")

def make():
    g = M.G3d(96)
    sd = R.seeded_state_dict(R.g3d_shapes(96), 81, prefix="G3d.")
    g.load_state_dict({k[len("G3d."):]: v for k, v in sd.items()})
    return g.to(dev).train()

x = R.seeded_tensor((1, 96, 8, 16, 16), 82).to(dev)
tgt = R.seeded_tensor((1, 96, 8, 16, 16), 83).to(dev)
loss_fn = lambda m, x: F.mse_loss(m(x), tgt)
eager, graphed, eager2 = make(), make(), make()
opt_e = torch.optim.SGD(eager.parameters(), lr=1e-2, momentum=0.9)
opt_g = torch.optim.SGD(graphed.parameters(), lr=1e-2, momentum=0.9)
opt_e2 = torch.optim.SGD(eager2.parameters(), lr=1e-2, momentum=0.9)
ref = [training.train_step(eager2, loss_fn, opt_e2, {"x": x}).item() for _ in range(7)]
print("eager-only reference :", " ".join(f"{v:.8f}" for v in ref))
step = training.GraphedTrainStep(graphed, loss_fn, opt_g, {"x": x}, warmup=2)
out = []
for i in range(3):
    le = training.train_step(eager, loss_fn, opt_e, {"x": x}); lg = step(x=x); out.append((le.item(), lg.item()))
for i in range(4):
    lg = step(x=x); le = training.train_step(eager, loss_fn, opt_e, {"x": x}); out.append((le.item(), lg.item()))
print("eager (interleaved)  :", " ".join(f"{a:.8f}" for a, b in out))
print("graph (interleaved)  :", " ".join(f"{b:.8f}" for a, b in out))
