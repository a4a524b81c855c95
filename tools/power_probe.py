"""Samples GPU clock / power (rocm-smi) while the dominant conv runs back to back: is the MFMA kernel power-limited?
usage: python tools/power_probe.py [seconds] [precision] [nomfma_lib]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import ops, _lib

_lib.load()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
x = torch.randn(8, 96, 16, 64, 64, device=dev)
pc = ops.PackedConv(torch.randn(96, 96, 3, 3, 3, device=dev) * 0.02, torch.randn(96, device=dev))
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().splitlines()))
        except Exception as e:
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.2)


def idle():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True, text=True).stdout
    print("idle:", out.strip().splitlines()[:3])


idle()
for _ in range(3):
    ops.conv3d(x, pc, precision=prec)
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv3d(x, pc, precision=prec)
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
print(f"precision {prec}: {n} launches, {e0.elapsed_time(e1) / n:.3f} ms per launch")
hdr = None
for t, lines in samples[:: max(1, len(samples) // 8)]:
    if len(lines) >= 2:
        if hdr is None:
            hdr = lines[0]
            print(hdr)
        print(f"t+{t - t0:4.1f}s", lines[1])
