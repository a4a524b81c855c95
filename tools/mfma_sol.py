"""Sustained rate of the f16x3 convs' MFMA stream at the package's power limit (VERDICT r3 #1a): runs mphip_debug_mfma_sol back to
back for `seconds` and samples rocm-smi.  usage: python tools/mfma_sol.py [seconds] [mode]   (mode 1: fragments read from LDS)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import _lib


def smi_sample():
    """(package W, sclk MHz) from rocm-smi, or (None, None)."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
        lines = [l for l in out.strip().splitlines() if l.strip()]
        hdr, row = lines[0].split(","), lines[1].split(",")
        w = sclk = None
        for h, v in zip(hdr, row):
            hl = h.lower()
            if "power" in hl and w is None:
                try:
                    w = float(v)
                except ValueError:
                    pass
            if "sclk" in hl and sclk is None:
                import re
                m = re.search(r"(\d+)\s*mhz", v.lower())
                if m:
                    sclk = float(m.group(1))
        return w, sclk
    except Exception:
        return None, None


def measure(seconds=2.0, mode=1, workgroups=256, iters=400):
    """-> dict(tflops_issued, power_w, sclk_mhz, seconds, launches): the MFMA stream alone, back to back."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sink = torch.empty(workgroups * 512, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((time.time(),) + smi_sample())
            time.sleep(0.1)

    def launch(n):
        for _ in range(n):
            _lib.check(lib.mphip_debug_mfma_sol(sink.data_ptr(), workgroups, iters, mode, stream), "mphip_debug_mfma_sol")

    launch(2)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        launch(10)
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1)
    mfmas = float(workgroups) * 8 * iters * 162 * n
    late = [s for s in samples if s[0] - t0 > 0.5 * seconds and s[1] is not None]   # after the package has settled at its limit
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    return {"tflops_issued": mfmas * 32768 / (ms * 1e-3) / 1e12, "power_w": med([s[1] for s in late]),
            "sclk_mhz": med([s[2] for s in late if s[2] is not None]), "seconds": ms * 1e-3, "launches": n, "mode": mode,
            "mfma_per_launch": float(workgroups) * 8 * iters * 162}


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    modes = [int(sys.argv[2])] if len(sys.argv) > 2 else [0, 1]
    for m in modes:
        r = measure(secs, m)
        print(f"mfma_sol mode {m} ({'fragments from LDS' if m else 'fragments in registers'}): {r['tflops_issued']:.0f} TFLOP/s issued "
              f"(= {r['tflops_issued'] / 3:.0f} algorithmic for the 3-product f16x3 arithmetic), {r['power_w']} W, sclk {r['sclk_mhz']} MHz, "
              f"{r['seconds']:.2f} s, {r['launches']} launches", flush=True)
