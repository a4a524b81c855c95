"""Energy of a whole hot-slice step: samples the package power (rocm-smi) while the B=8 inference loop runs back to back and
reports average power x time per step.  usage: python tools/power_step.py [seconds] [batch]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import model as M
from oracle import hotpath_ref as R

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
hot = M.GbaseHotSlice()
M.load_hot_state_dict(hot, R.seeded_gbase_hot_state_dict(7))
hot = hot.to(dev).eval()
inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(B, 21).items()}
samples, stop = [], False


def watts(csv_lines):
    hdr, row = csv_lines[0].split(","), csv_lines[1].split(",")
    for h, v in zip(hdr, row):
        if "ower" in h and "ax" not in h:
            try:
                return float(v)
            except ValueError:
                pass
    return None


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
        lines = out.strip().splitlines()
        if len(lines) >= 2:
            samples.append((time.time(), watts(lines), lines[1]))
        time.sleep(0.1)


with torch.no_grad():
    for _ in range(10):
        hot(**inp)
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
with torch.no_grad():
    while time.time() - t0 < secs:
        for _ in range(40):
            hot(**inp)
        n += 40
        torch.cuda.synchronize()
t1 = time.time()
stop = True
th.join()
ms = (t1 - t0) / n * 1e3
w = [s[1] for s in samples if s[1] is not None and s[0] - t0 > 1.0]
avg = sum(w) / max(1, len(w))
print(f"B={B}: {n} steps, {ms:.3f} ms/step, {B / ms * 1e3:.0f} frames/s; package power over {len(w)} samples: avg {avg:.0f} W, min {min(w):.0f}, max {max(w):.0f}")
print(f"energy per step {avg * ms / 1e3:.2f} J = {avg * ms / 1e3 / B:.3f} J per frame; at the 1400 W limit that energy takes {avg * ms / 1400:.2f} ms "
      f"({B / (avg * ms / 1400) * 1e3:.0f} frames/s): the step runs at {avg / 1400 * 100:.0f} % of the package limit on average")
for s in samples[:: max(1, len(samples) // 6)]:
    print(f"t+{s[0] - t0:4.1f}s", s[2])
