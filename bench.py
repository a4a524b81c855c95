#!/usr/bin/env python
"""bench.py — Gbase hot-slice throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot slice (reference model.py:1151-1171: S2C field -> 3D warp ->
G3d -> C2D field -> 3D warp -> depth sum) over one batch of synthetic 512x512 frames, i.e.
B x [96,16,64,64] appearance volumes already resident in HBM.  Frames are independent, so N GPUs
run N independent shards (weak scaling, no data-path collective); rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — the dominant kernel (conv3d 3x3x3 96->96 @16x64x64, 60 % of hot-slice FLOPs):
                 algorithmic FLOPs per launch / its average launch duration, measured live with
                 HIP events on the launch stream during the timed steps, vs the dense f16-MFMA peak,
                 and `sustained_peak`: the same MFMA stream alone, >= 2 s back to back, with package
                 power and clock (what the part sustains on this arithmetic at its power limit).
  cpu_baseline — the CPU oracle (oracle/hotpath_ref.py, ATen CPU ops = what the reference runs on a
                 CPU host) timed on this host on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# GPU_MAX_HW_QUEUES is left at the ROCm runtime's default (4).  r03-r05 raised it to 8 here (every stream of two batches in flight on its
# own hardware queue: +1.3 % on the headline, re-measured in r06: 3026 vs 3068 frames/s).  It is also what made `end_to_end_autocast_fp16`
# read 71 instead of 118 frames/s in r05: with more queues than the device keeps resident for the process, a side stream that lands on
# a queue which goes idle between steps pays a scheduling delay on every dispatch — 28 ms per Gbase.forward (profiles/NOTES_r06.md §1).
# `--hw-queues N` / an explicit environment setting still selects it (before the HIP runtime starts).
if "--hw-queues" in sys.argv[:-1]:
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(sys.argv[sys.argv.index("--hw-queues") + 1]))

import torch

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X dense f16/bf16 matrix peak (MI355X_MICROARCH.md)
FRAME_FLOPS = 164.1e9          # G3d 163.11 + 2 x FlowField 0.50 GFLOP per frame
FRAME_BYTES = 309e6            # layer-wise-minimal HBM bytes per frame
FINAL_CONV_FLOPS = 2.0 * 16 * 64 * 64 * 96 * 96 * 27   # G3d.final_conv per frame (32.6 GFLOP): skipped where nobody reads it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process (0: the ROCm runtime's default of 4; 8 gives each stream of two batches in flight "
                         "its own queue, +1.3 %% on the headline, at the price described at the top of this file)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (BASELINE config 2: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=0, help="1 = replay the step as one hipGraph, 0 = eager launches (default)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="independent batches kept in flight, each on its own HIP stream with its own C-side plan (side stream, "
                         "events, workspace): the latency-bound head and demand-driven tail of one step (few CUs busy) run beside "
                         "the other batch's work.  The K timed steps are issued round-robin over the streams; the line's "
                         "`one_in_flight` leg is the same loop on ONE stream (1 = only that)")
    ap.add_argument("--precision", default=None, choices=["fp32", "f16x3", "auto"],
                    help="conv arithmetic: fp32 = exact fp32 MFMA; f16x3/auto = split-f16 (3 f16 MFMAs per product, fp32-class "
                         "accuracy) where supported (default: MPHIP_CONV_PRECISION or auto)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames for the CPU baseline sample (0 = auto, ~15 s)")
    ap.add_argument("--torch-gpu-baseline", action="store_true",
                    help="also time the same graph through PyTorch-ROCm eager ops (ATen/MIOpen) on this GPU — what the "
                         "reference's own model.py would run on an MI355X — and add it as `torch_rocm_baseline` (off by "
                         "default: MIOpen's first-run kernel search can take minutes)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer (default) = the graded metric, BASELINE config 2; train = one training step of the hot slice "
                         "(forward + backward + SGD, BASELINE config 3's per-GPU shard: --batch 4) with the RCCL gradient "
                         "all-reduce when --gpus > 1 — a side measurement, separate JSON line")
    ap.add_argument("--single-stream-plan", action="store_true",
                    help="dev: plans without a side stream (MPHIP_PLAN_SINGLE_STREAM): both generator chains on the caller's stream, no "
                         "cross-stream event anywhere in a step")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no torchrun environment: print the torch.distributed.run command the "
                         "self-launcher would execute (one JSON line) and exit")
    ap.add_argument("--stub-worker", action="store_true",
                    help="launcher self-test (CPU, gloo): every rank joins the process group, rank 0 prints "
                         '{"stub_worker": true, "ranks": N}; no GPU work')
    ap.add_argument("--reenact-leg", action="store_true",
                    help="also time BASELINE config 5's per-GPU shard in its serving configuration (channels_last + MIOpen find mode; "
                         "the default line carries the immediate-mode leg `reenact_1x64`)")
    ap.add_argument("--e2e-nhwc", action="store_true", help="also time the end-to-end generator with channels_last 2D modules + MIOpen find mode")
    ap.add_argument("--extras-budget", type=float, default=100.0, help="seconds of side measurements after which remaining legs are skipped")
    ap.add_argument("--per-op", action="store_true", help="drive the step through the per-op Python schedule instead of the C-side plan")
    ap.add_argument("--full-final-conv", action="store_true",
                    help="evaluate G3d's last upsample + conv on every voxel (default: demand-driven, only what the final warp reads)")
    ap.add_argument("--repeats", type=int, default=5, help="extra K-step blocks timed after the graded one (their ms/step: `repeats` on the line)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements on the line (fp32_exact, roofline_hbm, end_to_end)")
    return ap.parse_args()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: re-executes itself under torch.distributed.run
    (one rank per GPU, rendezvous on 127.0.0.1), exactly the command the driver uses for N > 1.  Rank 0's JSON line
    passes through on stdout; the exit code is the launcher's."""
    import subprocess

    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    argv = [a for a in sys.argv[1:] if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    if not args.stub_worker:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def pin_rank(local_rank: int, world: int) -> dict:
    """One rank per GPU: give every rank its own contiguous slice of the host cores this process may use (Linux numbers the
    cores of a NUMA node contiguously, and GPUs are attached to nodes in order on the 8-GPU MI355X boards), and size the OpenMP /
    ATen thread pools to it — otherwise N ranks each start a pool as wide as the whole machine.  MPHIP_PIN_RANKS=0 disables."""
    if world <= 1 or os.environ.get("MPHIP_PIN_RANKS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {}
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {}
    torch.set_num_threads(max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", len(mine))))))
    return {"cores": [mine[0], mine[-1]], "threads": torch.get_num_threads()}


def stub_worker(args, rank, world):
    """CPU self-test of the launch path: rendezvous + one all-reduce over gloo, rank 0 reports."""
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"stub_worker": True, "ranks": dist.get_world_size(), "n_gpus": args.gpus, "sum": t.item()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


class DominantKernelTimer:
    """HIP events around every launch of the dominant conv during the timed region (events are
    recorded on torch's current stream, which is the stream ops.conv3d launches on)."""

    def __init__(self, shape):
        self.shape = shape
        self.events = []
        self.active = False

    def __call__(self, x, pc, launch):
        if self.active and tuple(x.shape[1:]) == self.shape[1:] and pc.co == self.shape[0] and pc.k == 3:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = launch()
            e1.record()
            self.events.append((e0, e1))
            return y
        return launch()

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))


def _sha256(path):
    import hashlib

    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_record(name, sources):
    """A committed rocprofv3 --pmc summary (profiles/<name>) — used ONLY when it is stamped with the sha256 of the kernel sources it
    measured and those are the sources of this build (tools/collect_profiles.sh stamps them); otherwise (None, True): the line then
    says `"traffic": null, "stale": true` instead of quoting counters of a kernel that has changed since."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            rec = json.load(f)
    except Exception:
        return None, True
    stamped = rec.get("_sources") or {}
    for src in sources:
        if stamped.get(src) != _sha256(os.path.join(ROOT, "megaportrait-hack_amd", "csrc", src)):
            return None, True
    return rec, False


def pmc_traffic(wino):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes, or (None, stale)."""
    rec, stale = pmc_record("r06_pmc_conv.json", ["conv3d_f16x3_wino_pp.hip", "mphip_wino_tile.h", "mphip_f16x3.h"] if wino else ["conv3d_f16x3.hip", "mphip_f16x3.h"])
    if rec is None:
        return None, stale
    try:
        return rec["derived"]["traffic_bytes_per_launch"], False
    except Exception:
        return None, True


def sustained_peak(seconds=2.2):
    """The f16x3 convs' MFMA stream alone (mphip_debug_mfma_sol, fragments read from LDS like the conv), back to back for `seconds`:
    issued TFLOP/s, package power, clock — the ceiling a perfectly overlapped conv kernel could reach on this arithmetic."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_sol

    r = mfma_sol.measure(seconds, mode=1)
    return {"tflops_issued": round(r["tflops_issued"], 1), "power_w": r["power_w"], "sclk_mhz": r["sclk_mhz"], "seconds": round(r["seconds"], 2),
            "what": "v_mfma_f32_32x32x16_f16 in the convs' three-products-per-tap order on random hi/lo fragments read from LDS (10 ds_read_b128 per "
                    "18 MFMAs), 3 x 2 accumulator tiles per wave, 2 waves per SIMD, every CU, no global traffic, no barriers (csrc/mfma_sol.hip)"}


def torch_rocm_baseline(dev, B, steps=5):
    """The reference-equivalent graph (oracle/hotpath_ref.py = the reference's module graph over ATen ops) run on
    the GPU through PyTorch-ROCm eager kernels (MIOpen conv3d, ATen grid_sample / group_norm / interpolate).
    A measured-beside baseline like cpu_baseline: never part of the product path."""
    from oracle import hotpath_ref as R

    sd = {k: v.to(dev) for k, v in R.seeded_gbase_hot_state_dict(7).items()}
    g = torch.Generator(device="cpu").manual_seed(20240501)
    inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g),
               zs=torch.randn(B, 512, generator=g), zd=torch.randn(B, 512, generator=g),
               Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
               ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    orig_eye, orig_linspace, orig_tensor = torch.eye, torch.linspace, torch.tensor
    # the restatement builds a few tiny constants on the default (CPU) device: route them to the GPU here
    torch.eye = lambda *a, **k: orig_eye(*a, **{**k, "device": k.get("device", dev)})
    torch.linspace = lambda *a, **k: orig_linspace(*a, **{**k, "device": k.get("device", dev)})
    torch.tensor = lambda *a, **k: orig_tensor(*a, **{**k, "device": k.get("device", dev)})
    try:
        with torch.no_grad():
            for _ in range(2):
                R.hot_slice(sd=sd, **inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                R.hot_slice(sd=sd, **inp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    finally:
        torch.eye, torch.linspace, torch.tensor = orig_eye, orig_linspace, orig_tensor
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 2),
            "kind": "PyTorch-ROCm eager (ATen/MIOpen fp32) of the same graph on the same GPU", "batch": B, "steps": steps}


def cpu_baseline(frames):
    """Times the CPU oracle on this host: B=1 frames through oracle.hotpath_ref.hot_slice, at the ATen thread count that is fastest
    on this host among {8, 16, 32, 64, all} (every core of a many-core host oversubscribes a B=1 problem)."""
    from oracle import hotpath_ref as R

    torch.manual_seed(20240501)
    sd = R.seeded_gbase_hot_state_dict(7)
    inp = R.seeded_hot_inputs(1, 3)
    all_threads = torch.get_num_threads()
    tried = {}
    with torch.no_grad():
        R.hot_slice(sd=sd, **inp)                      # warm-up (allocator, oneDNN primitives)
        for n in sorted({n for n in (8, 16, 32, 64, all_threads) if n <= all_threads}):
            torch.set_num_threads(n)
            R.hot_slice(sd=sd, **inp)
            t0 = time.perf_counter()
            R.hot_slice(sd=sd, **inp)
            tried[n] = 1.0 / (time.perf_counter() - t0)
        threads = max(tried, key=tried.get)
        torch.set_num_threads(threads)
        if frames <= 0:
            frames = max(2, min(40, int(12.0 * tried[threads])))
        t0 = time.perf_counter()
        for _ in range(frames):
            R.hot_slice(sd=sd, **inp)
        dt = time.perf_counter() - t0
        torch.set_num_threads(all_threads)
    return {"value": frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "threads_tried_frames_per_s": {str(k): round(v, 2) for k, v in tried.items()},
            "sample": f"{frames} frames (B=1, 96x16x64x64) through oracle/hotpath_ref.py hot_slice "
                      f"(ATen CPU fp32, {threads} threads = the fastest of {sorted(tried)} on this host), {dt:.1f} s"}


def end_to_end(dev, B, steps=8, warmup=3, fp16=False, channels_last=False):
    """Gbase.forward(xs, xd) -> (image, pyramids) on synthetic 512x512 RGB pairs (xs, xd ~ U[0,1)): the
    full generator — Eapp / Emtn / G2d bodies on PyTorch-ROCm (MIOpen), the 3D tail, the hot slice and G2d's head on
    the HIP kernels.  Random-init weights (gbase.Gbase builds offline).  Side measurement: the hot slice is ~10 % of
    the generator's FLOPs, the 2D convs dominate."""
    from megaportrait_hack_amd import gbase

    torch.manual_seed(20240501)
    g = gbase.Gbase().to(dev).eval()
    find_mode = torch.backends.cudnn.benchmark
    if channels_last:
        torch.backends.cudnn.benchmark = True   # MIOpen find mode: NHWC only pays off with it (gbase.Gbase.channels_last_2d)
        g.channels_last_2d()
    gen = torch.Generator(device="cpu").manual_seed(20240501)
    xs = torch.rand(B, 3, 512, 512, generator=gen).to(dev)
    xd = torch.rand(B, 3, 512, 512, generator=gen).to(dev)
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16, enabled=fp16):
        for _ in range(warmup):
            img, pyr = g(xs, xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            img, pyr = g(xs, xd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert img.shape == (B, 3, 512, 512) and torch.isfinite(img.float()).all()
    torch.backends.cudnn.benchmark = find_mode
    del g
    torch.cuda.empty_cache()
    prec = ("torch.autocast(float16) around the generator like the reference's training loop (train.py:188): MIOpen fp16 2D convs; under the "
            "same region the hot slice's F(2,3) convs issue ONE f16 product per multiply (model._autocast_policy; Eapp's 3-D tail, called "
            "block by block, keeps f16x3), fp32 everything else") if fp16 else "MIOpen fp32 2D convs"
    if channels_last:
        prec += "; motionEncoder and G2d in torch.channels_last (Gbase.channels_last_2d) with MIOpen find mode (cudnn.benchmark)"
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 2), "batch": B, "steps": steps,
            "workload": "gbase.Gbase.forward(xs, xd) -> (image [B,3,512,512], pyramids), xs/xd ~ U[0,1), random init; "
                        "2D encoders/decoder on PyTorch-ROCm, 3D tail + hot slice + G2d head on libmphip", "precision_2d": prec}


def reenact_leg(dev, drivers=64, chunk=16, repeats=3, find=True):
    """BASELINE config 5 on one GPU: ONE source image x `drivers` driver frames through gbase.Gbase.reenact — the source-side
    half (Eapp, Emtn(xs), S2C field, warp #1, G3d) runs once, per driver only Emtn(xd), the C2D field, the fused warp + depth sum
    and G2d; autocast-fp16 2D modules.  find=True: channels_last + MIOpen's find mode (the serving configuration; its kernel
    search adds ~2 minutes to a fresh process); find=False (the default line's compact leg): MIOpen immediate mode, NCHW."""
    from megaportrait_hack_amd import gbase

    torch.manual_seed(20240501)
    find_mode = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(find)
    g = gbase.Gbase().to(dev).eval()
    if find:
        g.channels_last_2d()
    gen = torch.Generator(device="cpu").manual_seed(20240502)
    xs = torch.rand(1, 3, 512, 512, generator=gen).to(dev)
    xd = torch.rand(drivers, 3, 512, 512, generator=gen).to(dev)
    out = g.reenact(xs, xd, chunk=chunk, fp16=True)   # warm-up (MIOpen find)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        out = g.reenact(xs, xd, chunk=chunk, fp16=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / repeats
    assert out.shape == (drivers, 3, 512, 512) and torch.isfinite(out).all()
    torch.backends.cudnn.benchmark = find_mode
    del g
    torch.cuda.empty_cache()
    return {"value": round(drivers / dt, 1), "unit": "driver frames/s", "ms_per_call": round(dt * 1e3, 1), "drivers": drivers, "chunk": chunk,
            "workload": "gbase.Gbase.reenact: 1 source x N drivers (BASELINE config 5's per-GPU shard), source-side half once per call; "
                        "autocast-fp16 2D modules (" + ("channels_last, MIOpen find mode" if find else "NCHW, MIOpen immediate mode") +
                        "), HIP kernels fp32 / f16x3"}


def roofline_hbm(hot, inp, B):
    """The two HBM-bound kernels of the slice in isolation: K2 = apply_warping_field
    (coords + gather), K3 = apply_warping_field + sum(dim=2), timed with HIP events on the launch stream for the model's own
    ("faithful") fields of this batch and for a smooth field that travels through the whole volume ("smooth": the
    stress case — the reference's fields only ever sample the 4^3 low corner).
    achieved = ALGORITHMIC bytes (K2 53.5 MB, K3 29.9 MB per frame) / time; `traffic` = counter bytes per launch from the
    committed rocprofv3 --pmc passes (profiles/r06_pmc_warps.json, B=8), `counter_GBps` = traffic / this run's time."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_warps as BW

    with torch.no_grad():
        w_s2c = hot.warp_generator_s2c(inp["Rs"], inp["ts"], inp["zs"], inp["es"])
    table = {"faithful": w_s2c, "smooth": BW.fields(B)["smooth"]}
    res = BW.measure(B, iters=20, quiet=True, field_override=table)
    rec_, stale = pmc_record("r06_pmc_warps.json", ["warp.hip"])   # (counters are quoted only when stamped with this build's warp.hip)
    pmc = rec_.get("kernels", {}) if rec_ else {}
    out = {}
    for key, rec in res.items():
        kname, kind = key.split(" / ")
        short = kname.split()[0]
        kernels = {"K2": ["warp_coords_kernel", "warp_corner_image_kernel", "warp_gather_kernel", "warp_gather_columns_kernel", "warp_gather_direct_kernel"],
                   "K3": ["warp_coords_kernel", "warp_gather_dsum_kernel"]}[short]
        traffic = None
        if B == 8:
            parts = [pmc.get(f"{k} / {kind} / B=8", {}).get("traffic_bytes_per_launch") for k in kernels]
            traffic = sum(p for p in parts if p) if any(parts) else None
        # `achieved` / `frac`: COUNTED HBM-side bytes per launch (the committed --pmc passes) / this run's launch time — what the
        # memory system actually moved (on the reference's fields the 25 MB/frame source read never happens, the
        # samples sit in a ~5^3 corner); the algorithmic figure (53.5 / 29.9 MB per frame) stays beside it
        counted = round(traffic / rec["ms"] / 1e6, 1) if traffic else None
        # (no counters stamped with this build's warp.hip -> no counted figure: the algorithmic one is NOT promoted in its place — on the
        #  reference's fields it counts a 25 MB/frame source read that never happens)
        out.setdefault(short, {})[kind] = {
            "kernels": kernels, "bound": "hbm", "launch_ms": rec["ms"], "achieved": counted,
            "peak": 8000.0, "unit": "GB/s", "frac": round(counted / 8000.0, 4) if counted else None,
            "accounting": "counted bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)" if counted else "no counters for this build (stale): see algorithmic_*",
            "traffic": traffic, "stale": bool(stale)}
        # The algorithmic figure (53.5 / 29.9 MB per frame over the launch time) is only meaningful where the samples really walk the source
        # volume — the smooth stress field.  On the model's own fields it would count a 25 MB/frame source read that never happens
        # (VERDICT r4 / r5): not printed there.
        if kind == "smooth":
            out[short][kind].update(algorithmic_GBps=rec["algorithmic_GBps"], algorithmic_frac=rec["frac_of_8TBps"])
    return out


def full_final_conv_leg(hot, inp, B, steps=20, warmup=3):
    """The same step with G3d's last upsample + conv evaluated on EVERY voxel (MPHIP_PLAN_FULL_FINAL_CONV) — what `one_in_flight`
    (steps on one stream) would be without the demand-driven tail; same output bits."""
    old = hot.full_final_conv
    hot.full_final_conv = True
    try:
        with torch.no_grad():
            for _ in range(warmup):
                hot(**inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                hot(**inp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    finally:
        hot.full_final_conv = old
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps}


def one_in_flight_leg(hot, inp, B, steps=20, warmup=3, peak=PEAK_F16_MFMA_TFLOPS):
    """The same K steps issued on ONE stream, each behind the previous one (r01-r02's headline loop), with the device-side spread of
    the steps and the dominant conv's launch time there (no other batch's kernels between its two events)."""
    plan = hot._plan_for(inp["vs"])
    with torch.no_grad():
        for _ in range(warmup):
            hot(**inp)
        torch.cuda.synchronize()
        plan.profile(True)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            hot(**inp)
            marks[i + 1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
    tot, cnt = plan.profile_read(0)
    plan.profile(False)
    dom_ms = tot / max(1, cnt)
    tf = 2.0 * B * 16 * 64 * 64 * 96 * 96 * 27 / (dom_ms * 1e-3) / 1e12 if cnt else None
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "step_ms": {"min": round(ms[0], 3), "median": round(ms[len(ms) // 2], 3), "max": round(ms[-1], 3)},
            "dominant_conv": ({"launch_ms": round(dom_ms, 4), "launches_timed": cnt, "achieved": round(tf, 2), "unit": "TFLOP/s",
                               "peak": peak, "frac": round(tf / peak, 4)} if cnt else None)}


def fp32_exact(hot, inp, B, steps=10, warmup=2):
    """The same hot slice with every conv on the exact fp32 MFMA kernels (precision 0) — beside the default f16x3 line."""
    from megaportrait_hack_amd import ops

    old = ops.get_conv_precision()
    ops.set_conv_precision("fp32")
    try:
        with torch.no_grad():
            for _ in range(warmup):
                hot(**inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                hot(**inp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    finally:
        ops.set_conv_precision(old)
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "dtype": "f32 (v_mfma_f32_32x32x2_f32 everywhere)"}


def eapp_tail_leg(dev, B=8, steps=10, warmup=3):
    """Scope row f1: Eapp's 3-D tail (model.py:217-226, 271-290) — six ResBlock3D_Adaptive(96, 96) = twelve full-resolution
    96->96 3x3x3 convs per frame on the appearance volume, the heaviest source-side piece of the generator — alone, random-init weights,
    input resident in HBM.  Side measurement."""
    from megaportrait_hack_amd import model as M

    torch.manual_seed(0)
    tail = M.Eapp3DTail().to(dev).eval()
    x = torch.randn(B, 1536, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(warmup):
            tail(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tail(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    flops = 12 * 2.0 * B * 65536 * 96 * 96 * 27
    return {"value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "batch": B, "steps": steps,
            "algorithmic_tflops": round(flops * steps / dt / 1e12, 1),
            "workload": "model.Eapp3DTail.forward on [B,1536,64,64] (= [B,96,16,64,64]): 6 x ResBlock3D_Adaptive(96,96), f16x3 F(2,3) convs, "
                        "GroupNorm statistics from the conv epilogues"}


def train_leg(dev, B=4, steps=10, warmup=3, autocast=False):
    """BASELINE config 3's per-GPU shard on the default line: one training step of the hot slice (forward + backward + SGD,
    eager launches) at B=4, 96x16x64x64.  Side measurement; `--mode train` is the full-featured version (hipGraph, N ranks).
    autocast: the forward runs inside torch.autocast(float16) like the reference's generator step (train.py:145,188) — the F(2,3) convs
    (forward and bwd-data) and the 3x3x3 bwd-weight kernel then use one f16 product per multiply (include/mphip.h: mphip_conv3d_set_half_products)."""
    import torch.nn.functional as F

    from megaportrait_hack_amd import model as M, training

    torch.manual_seed(20240501)
    hot = M.GbaseHotSlice().to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(20240501)
    inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g),
               zs=torch.randn(B, 512, generator=g), zd=torch.randn(B, 512, generator=g),
               Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
               ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    inp["vs"].requires_grad_(True)
    tgt = torch.randn(B, 96, 64, 64, generator=g).to(dev)
    if autocast:
        def loss_fn(m, **kw):
            with torch.autocast("cuda", dtype=torch.float16):
                out = m(**kw)
            return F.mse_loss(out.float(), tgt)
    else:
        loss_fn = lambda m, **kw: F.mse_loss(m(**kw), tgt)
    opt = torch.optim.SGD(hot.parameters(), lr=1e-5)
    from megaportrait_hack_amd import ops as _ops
    table = None
    for _ in range(warmup):
        loss = training.train_step(hot, loss_fn, opt, inp, pack_table=table)
        table = table or _ops.PackTable.from_module(hot)   # (after the first step: every later one re-packs in <= 5 launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = training.train_step(hot, loss_fn, opt, inp, pack_table=table)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).all()
    # the same step replayed as ONE hipGraph (training.GraphedTrainStep): the eager step is host-bound (~930 launches)
    graphed = training.GraphedTrainStep(hot, loss_fn, opt, inp, warmup=2)
    for _ in range(2):
        gl = graphed(**inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gl = graphed(**inp)
    torch.cuda.synchronize()
    gdt = time.perf_counter() - t0
    assert torch.isfinite(gl).all()
    del hot, opt, graphed
    torch.cuda.empty_cache()
    return {"value": round(B * steps / gdt, 2), "unit": "frames/s", "ms_per_step": round(gdt / steps * 1e3, 3), "batch": B, "steps": steps,
            "launch": "one hipGraph per step (training.GraphedTrainStep)", "eager_ms_per_step": round(dt / steps * 1e3, 3),
            "workload": "GbaseHotSlice training step (forward + backward + SGD), BASELINE config 3's per-GPU shard; final_conv forward and "
                        "backward demand-driven (the final warp's sample boxes)",
            "dtype": ("autocast(float16) policy: one f16 product per multiply (fp32 accumulate) in the F(2,3) convs (forward, bwd-data) and the 3x3x3 "
                      "bwd-weight kernel, f16x3 in the other convs, fp32 everything else") if autocast else "f16x3 forward/backward convs, fp32 everything else"}


def train_mode(args, rank, world, dev, dist):
    """Side measurement (not the graded metric): training step of the hot slice, frames sharded over ranks, gradients
    averaged with bucketed RCCL all-reduces (training.train_step); --graph 1 replays the step as one hipGraph (1 GPU)."""
    import torch.nn.functional as F

    from megaportrait_hack_amd import _lib, model as M, training

    _lib.load()
    B = args.batch if args.batch != 8 else 4   # config 3: batch 32 over 8 GPUs
    torch.manual_seed(20240501)                 # identical replicas on every rank
    hot = M.GbaseHotSlice().to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(20240501 + rank)
    inp = dict(vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g),
               zs=torch.randn(B, 512, generator=g), zd=torch.randn(B, 512, generator=g),
               Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
               ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    inp["vs"].requires_grad_(True)              # vs comes from Eapp: its gradient is part of the step
    tgt = torch.randn(B, 96, 64, 64, generator=g).to(dev)
    loss_fn = lambda m, **kw: F.mse_loss(m(**kw), tgt)
    opt = torch.optim.SGD(hot.parameters(), lr=1e-5)
    if args.graph and world == 1:
        graphed = training.GraphedTrainStep(hot, loss_fn, opt, inp)
        step = lambda: graphed(**inp)
    else:
        # N > 1: bucketed all-reduces launched from gradient hooks, underneath backward, on in-place flat gradient buffers
        reducer = training.OverlappedGradReducer(hot.parameters()) if world > 1 else None
        training.train_step(hot, loss_fn, opt, inp, reducer=reducer)      # one real step creates the packs ...
        from megaportrait_hack_amd import ops as _ops
        table = _ops.PackTable.from_module(hot)                           # ... which every later step re-makes in <= 5 launches
        step = lambda: training.train_step(hot, loss_fn, opt, inp, reducer=reducer, pack_table=table)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync_all()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).all()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        n_param = sum(p.numel() for p in hot.parameters())
        print(json.dumps({
            "metric": "hot-slice training step frames/sec (fwd + bwd + SGD; BASELINE config 3 shard)", "value": round(world * B * args.steps / dt, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 forward/backward convs, fp32 everything else", "data": "synthetic",
            "config": {"workload": "GbaseHotSlice train step (model.py:1151-1171 under autograd)", "frames_per_gpu": B,
                       "global_batch": B * world, "parallelism": f"dp{world} (frame shards, gradient all-reduce: "
                       f"{n_param * 4 / 1e6:.0f} MB fp32 per step)" if world > 1 else "single GPU",
                       "hip_graph": bool(args.graph and world == 1), "optimizer": "SGD"}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # self-launch: N ranks under torch.distributed.run
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.stub_worker:
        return stub_worker(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    pinned = pin_rank(local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
    if args.mode == "train":
        return train_mode(args, rank, world, dev, dist)

    from megaportrait_hack_amd import _lib, model as M, ops

    _lib.load()  # fail loudly if libmphip.so is missing
    if args.precision:
        ops.set_conv_precision(args.precision)
    f16x3 = ops.get_conv_precision() == 1
    B = args.batch
    torch.manual_seed(20240501 + rank)
    hot = M.GbaseHotSlice().to(dev).eval()   # PyTorch default init (random weights; no checkpoints offline)
    g = torch.Generator(device="cpu").manual_seed(20240501 + rank)
    inp = dict(
        vs=torch.randn(B, 96, 16, 64, 64, generator=g), es=torch.randn(B, 512, generator=g),
        zs=torch.randn(B, 512, generator=g), zd=torch.randn(B, 512, generator=g),
        Rs=(torch.rand(B, 3, generator=g) * 60 - 30), Rd=(torch.rand(B, 3, generator=g) * 60 - 30),
        ts=torch.randn(B, 3, generator=g) * 0.1, td=torch.randn(B, 3, generator=g) * 0.1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    # the timed steps rotate over distinct input sets (different volumes, codes and head poses: the demand-driven tail's tile list, the
    # range descriptors and the warps' source boxes are recomputed from the data every step)
    g2 = torch.Generator(device="cpu").manual_seed(20240601 + rank)
    inp_b = dict(
        vs=torch.randn(B, 96, 16, 64, 64, generator=g2) * 1.3, es=torch.randn(B, 512, generator=g2),
        zs=torch.randn(B, 512, generator=g2), zd=torch.randn(B, 512, generator=g2),
        Rs=(torch.rand(B, 3, generator=g2) * 60 - 30), Rd=(torch.rand(B, 3, generator=g2) * 60 - 30),
        ts=torch.randn(B, 3, generator=g2) * 0.1, td=torch.randn(B, 3, generator=g2) * 0.1)
    inp_sets = [inp, {k: v.to(dev) for k, v in inp_b.items()}]

    dom = DominantKernelTimer((96, 96, 16, 64, 64))
    step = hot
    if args.single_stream_plan:
        hot.overlap_generators = False
    use_plan = bool(hot.use_c_plan and not args.graph and not args.per_op)
    if args.full_final_conv:
        hot.full_final_conv = True
    if use_plan:
        # default: ONE ctypes call per step into the C-side plan (csrc/plan.hip); the dominant conv is timed by HIP events the
        # plan records around its launches on the launch stream (a Python hook would force the per-op path)
        with torch.no_grad():
            hot(**inp)                      # builds the plan, packs the weights
        plan = hot._plan_for(inp["vs"])
    elif args.graph:
        # the dominant conv is timed with HIP events in a short eager pass (events cannot be queried inside
        # a captured graph); the throughput loop then replays the captured step.
        ops.set_conv_hook(dom)
        with torch.no_grad():
            for _ in range(2):
                hot(**inp)
            torch.cuda.synchronize()
            dom.active = True
            for _ in range(5):
                hot(**inp)
            torch.cuda.synchronize()
            dom.active = False
        ops.set_conv_hook(None)
        step = M.GraphedHotSlice(hot, inp)
        inp_sets = [inp]   # (a replayed graph reads its static input buffers)
    else:
        ops.set_conv_hook(dom)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        lanes = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.inflight))] if args.inflight > 1 and not args.graph else None
        if lanes is None:
            for _ in range(args.warmup):
                out = step(**inp)
        else:   # every lane's stream gets its own plan (side stream, events, packed weights): build and warm them outside the timed region
            for lane in lanes:
                lane.wait_stream(torch.cuda.current_stream())
            for i in range(max(args.warmup, 2 * len(lanes))):
                with torch.cuda.stream(lanes[i % len(lanes)]):
                    out = step(**inp)
            if use_plan:
                with torch.cuda.stream(lanes[0]):
                    plan = hot._plan_for(inp["vs"])   # (the dominant-conv events come from lane 0's plan)
        sync_all()
        dom.active = not args.graph
        if use_plan:
            plan.profile(True)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if lanes is None else None

        def timed_block(record_marks):
            """EXACTLY K steps between barrier + synchronize on both sides; returns the wall time."""
            sync_all()
            t0_ = time.perf_counter()
            out_ = None
            if lanes is None:
                if record_marks:
                    marks[0].record()
                for i in range(args.steps):
                    out_ = step(**inp_sets[i % len(inp_sets)])
                    if record_marks:
                        marks[i + 1].record()      # device-side step boundaries (an event record, no sync): the spread of the K steps
            else:
                for lane in lanes:
                    lane.wait_stream(torch.cuda.current_stream())
                for i in range(args.steps):
                    with torch.cuda.stream(lanes[i % len(lanes)]):
                        out_ = step(**inp_sets[(i // len(lanes)) % len(inp_sets)])
                for lane in lanes:
                    torch.cuda.current_stream().wait_stream(lane)
            sync_all()
            return time.perf_counter() - t0_, out_

        dt, out = timed_block(True)
        dom.active = False
        prof = None
        if use_plan:   # (the roofline events come from the K graded steps only: read them before the repeat blocks)
            prof = (plan.profile_read(0), plan.profile_read(1))
            plan.profile(False)
        # the same K-step block a few more times: `value` stays the first block, the spread is reported beside it
        rep_ms = []
        for _ in range(max(0, args.repeats)):
            rdt, _o = timed_block(False)
            rep_ms.append(rdt / args.steps * 1e3)
    assert torch.isfinite(out).all()

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rep_ms:
            t = torch.tensor(rep_ms, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rep_ms = [float(v) for v in t.tolist()]

    step_ms = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])) if marks else []
    dom_ms, dom_launches, demand = dom.mean_ms(), len(dom.events), None
    if use_plan:
        (tot, dom_launches), (dtot, dcnt) = prof
        dom_ms = tot / max(1, dom_launches)
        if dcnt:
            demand = {"launch_ms": round(dtot / dcnt, 4), "launches_timed": dcnt,
                      "what": "G3d's last conv (same kernel family, 4x8x8 tiles) evaluated only on the output tiles the final warp reads: "
                              "the sample boxes of apply_warping_field + sum(dim=2) (model.py:1167-1171) are known from the C2D field "
                              "before the conv runs; on the reference's fields ~2 of a frame's 512 tiles; bit-identical output"}
    dom_flops = 2.0 * B * 16 * 64 * 64 * 96 * 96 * 27
    line = None
    if rank == 0:
        fps = world * B * args.steps / dt
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        wino = bool(f16x3 and _lib.load().mphip_conv3d_kernel_variant(B, 96, 96, 16, 64, 64, 3, 1) == 5)
        issued_per_alg = 2.0 if wino else 3.0   # f16 MFMA FLOPs issued per algorithmic FLOP: 3 products, x 2/3 in the F(2,3) domain
        if f16x3:
            dom_name = (("conv3d_k3_f16x3_wino_pp_kernel (Conv3d 3x3x3 96->96 @16x64x64 in the 1-D Winograd F(2,3) domain: 2/3 of the direct "
                         "kernel's MFMAs, role-split schedule, " if wino else "conv3d_k3_f16x3_kernel<4,8,16,8,1> (Conv3d 3x3x3 96->96 @16x64x64, ")
                        + ("2 full launches/step + 1 demand-driven" if demand else "3 launches/step") + ")")
            peak, dtype = PEAK_F16_MFMA_TFLOPS, ("f16x3 (fp32 in/out, operands split into 2 f16 halves, 3 f16 MFMAs per product, fp32 accumulate"
                                                 + ("; 3x3x3 convs that fill the chip run in the F(2,3) Winograd domain along W" if wino else "") + ")")
        else:
            dom_name = "conv3d_k3_tiled_kernel<4,8,8,3,2> (Conv3d 3x3x3 96->96 @16x64x64, 3 launches/step)"
            peak, dtype = PEAK_F32_MFMA_TFLOPS, "f32"
        traffic, stale = pmc_traffic(wino) if (f16x3 and B == 8) else (None, False)
        f16_note = lambda ach: (
            f"algorithmic (fp32-equivalent) FLOPs of the reference's conv; the kernel issues {issued_per_alg:g}x that on the f16 pipe "
            f"(3 split products{', x 2/3 in the F(2,3) domain' if wino else ''}): {round(issued_per_alg * ach, 1)} TFLOP/s = "
            f"{round(issued_per_alg * ach / peak, 4)} of the f16 peak; vs the fp32-MFMA peak ({PEAK_F32_MFMA_TFLOPS}) the algorithmic rate is "
            f"{round(ach / PEAK_F32_MFMA_TFLOPS, 2)}x.  The 2500 TFLOP/s peak assumes 2.4 GHz; on dense f16 MFMAs the part clocks "
            "~1.65-1.75 GHz at its power limit: `sustained_peak` is the same MFMA stream alone, measured in this run")
        line = {
            "metric": "Gbase fwd hot-slice frames/sec @512^2 (96ch 16x64x64 volume)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "Gbase hot slice (model.py:1151-1171): WarpGeneratorS2C -> 3D warp -> G3d -> "
                                   "WarpGeneratorC2D -> 3D warp + depth sum; BASELINE config 2 (inference 512x512, "
                                   "96ch 16x64x64 volume), inputs resident in HBM, random-init weights",
                       "frames_per_gpu_per_step": B, "global_batch": world * B, "parallelism": f"dp{world} (frame shards, no collective)",
                       "launch": ("hipGraph replay of the captured step" if args.graph else
                                  ("one call per step into the C-side plan (mphip_hot_slice_forward), eager stream launches"
                                   + (f"; the K steps are issued round-robin over {len(lanes)} HIP streams, each with its own plan "
                                      "(independent batches in flight: `one_in_flight` is the same loop on one stream)" if lanes else ""))
                                  if use_plan else
                                  "eager stream launches, per-op Python schedule"),
                       "final_conv": ("demand-driven: only the tiles the final warp reads (mphip_conv3d_fwd_roi)" if demand else "evaluated everywhere"),
                       "batches_in_flight": 1 if (args.graph or args.inflight < 2) else args.inflight,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "rank0_host_pinning": pinned or None,
            "step_ms": ({"min": round(step_ms[0], 3), "median": round(step_ms[len(step_ms) // 2], 3), "max": round(step_ms[-1], 3)}
                        if step_ms else None),
            # the reference graph's FLOPs per frame x frames/s (NOT a machine rate: the demand-driven tail skips most of final_conv and
            # the F(2,3) kernels execute 2/3 of a conv's multiplies) ...
            "hot_slice_tflops_reference_graph": round(fps * FRAME_FLOPS / 1e12, 2),
            # ... and with the skipped part of final_conv subtracted (still the reference's multiply count for what is evaluated)
            "hot_slice_tflops_executed": round(fps * (FRAME_FLOPS - (FINAL_CONV_FLOPS if demand else 0.0)) / 1e12, 2),
            "hot_slice_vs_f32_mfma_peak": round(fps * FRAME_FLOPS / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "hot_slice_layerwise_GBps": round(fps * FRAME_BYTES / 1e9, 1),
            "roofline": {"kernel": dom_name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "stale": stale,
                         "launch_ms": round(dom_ms, 4), "launches_timed": dom_launches,
                         "demand_driven_launch": demand,
                         "flops_per_launch": dom_flops, "f16_flops_issued_per_algorithmic_flop": issued_per_alg if f16x3 else None,
                         "note": (f16_note(achieved)
                                  + ("; with several batches in flight the two events around a launch also see the time the launch "
                                     "waits for CUs another batch's kernel still holds: `one_in_flight.dominant_conv` is the same "
                                     "measurement with the steps on one stream" if lanes else "")) if f16x3 else
                                 "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)"},
            "repeats": ({"ms_per_step": [round(v, 3) for v in rep_ms], "min": round(min(rep_ms), 3), "median": round(sorted(rep_ms)[len(rep_ms) // 2], 3),
                         "max": round(max(rep_ms), 3), "what": f"{len(rep_ms)} more blocks of the same {args.steps} steps (barrier + synchronize on "
                         "both sides each), after the graded block; the steps of every block rotate over 2 distinct input sets"} if rep_ms else None),
        }
        if world == 1 and not args.no_extras:
            # side measurements (N=1 only, never inside the timed region): each leg is timed, and legs that would push the
            # default run past its budget are skipped and say so
            ops.set_conv_hook(None)
            t_legs, secs = time.perf_counter(), {}

            def leg(key, fn, *a, **kw):
                spent = time.perf_counter() - t_legs
                if spent > args.extras_budget:
                    line[key] = {"skipped": f"extras budget ({args.extras_budget:.0f} s) spent after {spent:.0f} s"}
                    return
                t0_ = time.perf_counter()
                line[key] = fn(*a, **kw)
                secs[key] = round(time.perf_counter() - t0_, 1)

            if lanes is not None and use_plan:
                leg("one_in_flight", one_in_flight_leg, hot, inp, B, peak=peak)
                dc = (line.get("one_in_flight") or {}).get("dominant_conv")
                if dc and f16x3:
                    # `value` comes from the two-batch loop, so the line's primary launch_ms / achieved / frac are the events of THAT loop
                    # (VERDICT r5 #8): between the two events of a launch sit the other batch's small kernels too — they hold CUs while
                    # some of the conv's persistent workgroups start, and the launch ends with its last one.  `isolated_one_stream`: the
                    # same kernel, process and events with the K steps on ONE stream (the `one_in_flight` leg) — the kernel by itself,
                    # and the number a kernel trace of this command shows (under rocprofv3 the host is too slow to keep two batches going).
                    r = line["roofline"]
                    r["in_two_batch_loop"] = {k: r[k] for k in ("launch_ms", "launches_timed", "achieved", "frac")}   # (= the primary fields)
                    r["isolated_one_stream"] = {"launch_ms": dc["launch_ms"], "launches_timed": dc["launches_timed"], "achieved": dc["achieved"],
                                                "frac": dc["frac"], "note": f16_note(dc["achieved"])}
                    r["measured"] = ("HIP events stamped with the kernel's begin / end (hipExtLaunchKernelGGL) during the timed region of the "
                                     "headline (two batches in flight); `isolated_one_stream`: the same events with K steps on ONE stream")
            if f16x3:
                leg("_sustained", sustained_peak)
                sp = line.pop("_sustained", None)
                if sp and "tflops_issued" in sp:
                    r = line["roofline"]
                    r["sustained_peak"] = sp
                    iso = r.get("isolated_one_stream") or r
                    r["frac_of_sustained"] = round(issued_per_alg * iso["achieved"] / sp["tflops_issued"], 4)
                    r["frac_of_sustained_what"] = ("f16 FLOP/s this kernel ISSUES (achieved x f16_flops_issued_per_algorithmic_flop) / the rate the same "
                                                   "MFMA stream sustains alone on this package in this run")
            if demand:
                leg("full_final_conv", full_final_conv_leg, hot, inp, B)
            leg("roofline_hbm", roofline_hbm, hot, inp, B)
            if f16x3:
                leg("fp32_exact", fp32_exact, hot, inp, B)
            leg("eapp_3d_tail", eapp_tail_leg, dev, B)                           # scope row f1: the source side's 3-D tail
            leg("train_step", train_leg, dev)                                    # BASELINE config 3's per-GPU shard
            leg("train_step_autocast", train_leg, dev, autocast=True)           # ... under the reference's autocast(float16) policy
            leg("reenact_1x64", reenact_leg, dev, repeats=2, find=False)         # BASELINE config 5's per-GPU shard
            leg("end_to_end", end_to_end, dev, B, steps=5, warmup=2)
            leg("end_to_end_autocast_fp16", end_to_end, dev, B, steps=5, warmup=2, fp16=True)
            if args.e2e_nhwc:      # opt-in: MIOpen's find pass adds ~1 minute
                line["end_to_end_autocast_fp16_nhwc"] = end_to_end(dev, B, fp16=True, channels_last=True)
            if args.reenact_leg:   # opt-in: MIOpen's find pass for its shapes adds ~2 minutes to the run
                line["reenact_1_source_x_64_drivers"] = reenact_leg(dev)
            line["leg_seconds"] = secs
        if world == 1 and args.torch_gpu_baseline:
            line["torch_rocm_baseline"] = torch_rocm_baseline(dev, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
