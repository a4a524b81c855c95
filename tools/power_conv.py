"""Package power / clock while ONE conv layer runs back to back (rocm-smi sampled every 0.1 s): is the kernel at the power limit, and what
clock does it get?  usage: python tools/power_conv.py [seconds] [B Ci Co D H W]   (MPHIP_LIB / MPHIP_WINOGRAD select the kernel)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MPHIP_ALLOW_ABLATED", "1")   # dev tool: may be pointed at a timing variant (csrc/mphip_ablate.h)
import torch
from megaportrait_hack_amd import ops, _lib
from mfma_sol import smi_sample

_lib.load()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
B, Ci, Co, D, H, W = (int(a) for a in sys.argv[2:8]) if len(sys.argv) >= 8 else (8, 96, 96, 16, 64, 64)
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, D, H, W, device=dev)
pc = ops.PackedConv(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02, torch.randn(Co, device=dev))
xr = ops.tensor_range(x)
for _ in range(3):
    ops.conv3d(x, pc, precision=1, x_range=xr)
torch.cuda.synchronize()
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        samples.append((time.time(),) + smi_sample())
        time.sleep(0.1)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv3d(x, pc, precision=1, x_range=xr)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop[0] = True; th.join()
late = [s for s in samples if s[0] - t0 > 0.5 * secs and s[1] is not None]
med = lambda v: sorted(v)[len(v) // 2] if v else None
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * D * H * W * Ci * Co * 27
print(f"{os.path.basename(os.environ.get('MPHIP_LIB', 'default')):28s} wino={os.environ.get('MPHIP_WINOGRAD', '1')}  {ms:.3f} ms/launch {fl / ms / 1e9:6.0f} TF/s alg  "
      f"{med([s[1] for s in late])} W  sclk {med([s[2] for s in late if s[2] is not None])} MHz", flush=True)
