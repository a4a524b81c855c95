"""G3d training step (forward + backward, SGD) through the HIP path vs the same nn graph in PyTorch-ROCm eager
on the same GPU.  Dev tool — not the graded bench (bench.py).   usage: bench_train.py [B] [--torch] [--iters n]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("B", nargs="?", type=int, default=4)
ap.add_argument("--torch", action="store_true", help="also time the PyTorch-ROCm eager graph")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", choices=["hip", "torch"], default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")


class TorchResBlock3D(nn.Module):  # the reference's graph (model.py:500-528) as plain nn modules
    def __init__(s, ci, co):
        super().__init__()
        s.conv1, s.gn1 = nn.Conv3d(ci, co, 3, padding=1), nn.GroupNorm(32, co)
        s.conv2, s.gn2 = nn.Conv3d(co, co, 3, padding=1), nn.GroupNorm(32, co)
        s.shortcut = nn.Conv3d(ci, co, 1) if ci != co else nn.Identity()

    def forward(s, x):
        i = s.shortcut(x)
        o = F.relu(s.gn1(s.conv1(x)))
        return F.relu(s.gn2(s.conv2(o)) + i)


class TorchG3d(nn.Module):
    def __init__(s):
        super().__init__()
        R, P = TorchResBlock3D, lambda: nn.AvgPool3d(2, 2)
        U = lambda: nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        s.downsampling = nn.Sequential(R(96, 96), P(), R(96, 192), P(), R(192, 384), P(), R(384, 768))
        s.upsampling = nn.Sequential(R(768, 384), U(), R(384, 192), U(), R(192, 96), U())
        s.final_conv = nn.Conv3d(96, 96, 3, padding=1)

    def forward(s, x):
        return s.final_conv(s.upsampling(s.downsampling(x)))


def step_time(model, x, tgt, iters):
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = F.mse_loss(model(x), tgt)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): loss = step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, loss.item()


x = torch.randn(a.B, 96, 16, 64, 64, device=dev)
tgt = torch.randn(a.B, 96, 16, 64, 64, device=dev)
tg = TorchG3d().to(dev)
if a.only != "torch":
    from megaportrait_hack_amd import model as M
    g = M.G3d(96).to(dev).train()
    g.load_state_dict(tg.state_dict())
    ms, loss = step_time(g, x, tgt, a.iters)
    print(f"HIP   G3d train step B={a.B}: {ms:8.2f} ms  ({a.B / ms * 1e3:.1f} frames/s)  loss {loss:.6f}")
if a.torch or a.only == "torch":
    ms, loss = step_time(tg, x, tgt, a.iters)
    print(f"torch G3d train step B={a.B}: {ms:8.2f} ms  ({a.B / ms * 1e3:.1f} frames/s)  loss {loss:.6f}")
