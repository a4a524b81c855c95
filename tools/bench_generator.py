"""dev: time of one warp generator (FlowField chain + compose) alone, B = 8 and 1, eager launches through the per-op path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M, ops

dev = torch.device("cuda:0")
torch.manual_seed(20240501)
hot = M.GbaseHotSlice().to(dev).eval()
for B in (8, 1):
    g = torch.Generator(device="cpu").manual_seed(1)
    R, t, z, e = (torch.rand(B, 3, generator=g) * 60 - 30).to(dev), (torch.randn(B, 3, generator=g) * 0.1).to(dev), torch.randn(B, 512, generator=g).to(dev), torch.randn(B, 512, generator=g).to(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.no_grad():
        for _ in range(3):
            w = hot.warp_generator_s2c(R, t, z, e)
        torch.cuda.synchronize()
        ops.begin_capture()
        with torch.cuda.graph(gr):
            w = hot.warp_generator_s2c(R, t, z, e)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gr.replay()
        torch.cuda.synchronize()
        print(f"B={B}: one generator, graph replay {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms  (checksum {float(w.double().sum()):.6f})")
