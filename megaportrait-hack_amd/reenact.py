"""Batch cross-reenactment CLI (BASELINE config 5): ONE source frame x N driver frames.

    python -m megaportrait_hack_amd.reenact --checkpoint Gbase.pth --source src.png --drivers d0.png d1.png ... \\
           --output-dir out/ [--gpus N]
    python -m megaportrait_hack_amd.reenact --config configs/inference/stage1-base.yaml        # the reference's yaml

The fixed equivalent of the reference's `inference.py` (inference.py:48-75): same checkpoint handling
(`load_state_dict(strict=False)` of a raw or wrapped checkpoint, inference.py:59-60), same pre/post-processing
(ToTensor + Normalize(0.5, 0.5) in, `(x + 1) / 2 * 255` out: inference.py:16-19,40-41), but it unpacks the
`(image, pyramids)` tuple the generator returns (the reference treats it as a tensor, inference.py:35-38, and crashes)
and runs many drivers per source: the source-side half of the graph (Eapp, S2C warp, G3d) is computed once
(`Gbase.reenact`).  With --gpus N the drivers are sharded by frame over N ranks (one process per GPU, no collective):
each rank writes its own frames.

Tensor I/O (`--source-tensor / --drivers-tensor / --output-tensor`, .pt or .npy, already normalised) bypasses the image
codecs; it is what the GPU test drives.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import List


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m megaportrait_hack_amd.reenact", description=__doc__.split("\n")[0])
    ap.add_argument("--config", help="reference-style yaml (inference.checkpoint_path / source_image / driving_image / output_image)")
    ap.add_argument("--checkpoint", help="Gbase checkpoint: raw state-dict (Gbase.pth) or a train.py checkpoint_epochN.pth")
    ap.add_argument("--source", help="source image file")
    ap.add_argument("--drivers", nargs="*", default=[], help="driver image files")
    ap.add_argument("--drivers-dir", help="directory of driver frames (sorted by name)")
    ap.add_argument("--source-tensor", help=".pt/.npy [1,3,H,W], already normalised")
    ap.add_argument("--drivers-tensor", help=".pt/.npy [N,3,H,W], already normalised")
    ap.add_argument("--output-dir", default=".", help="where frame_%%05d.png files go")
    ap.add_argument("--output-tensor", help="write the generated frames as one .pt tensor per rank instead of images")
    ap.add_argument("--chunk", type=int, default=16, help="driver frames per launch group")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--unit-range", action="store_true",
                    help="write image*255 (G2d ends in a sigmoid) instead of the reference's (x+1)/2*255 (inference.py:40)")
    ap.add_argument("--channels-last", action="store_true",
                    help="run motionEncoder and G2d in torch.channels_last (MIOpen NHWC kernels; +15 %% frames/s with --fp16 on MI355X)")
    ap.add_argument("--fp16", action="store_true",
                    help="run the PyTorch-ROCm 2D modules under torch.autocast(float16) (the reference's policy, train.py:188); "
                         "the HIP hot path stays fp32-class")
    ap.add_argument("--any-size", action="store_true", help="skip the reference's 512x512-only assert (model.py:1157)")
    ap.add_argument("--random-init", action="store_true", help="no checkpoint: random weights (plumbing tests)")
    ap.add_argument("--dry-run", action="store_true", help="resolve inputs and the launch plan, print them as JSON, exit")
    return ap.parse_args(argv)


def resolve(args) -> dict:
    """Argument / yaml plumbing (no torch): the job description the worker executes."""
    job = dict(checkpoint=args.checkpoint, source=args.source, drivers=list(args.drivers), source_tensor=args.source_tensor,
               drivers_tensor=args.drivers_tensor, output_dir=args.output_dir, output_tensor=args.output_tensor,
               output_files=None)
    if args.config:
        import yaml

        with open(args.config) as f:
            cfg = (yaml.safe_load(f) or {}).get("inference", {})
        job["checkpoint"] = job["checkpoint"] or cfg.get("checkpoint_path")
        job["source"] = job["source"] or cfg.get("source_image")
        if not job["drivers"] and cfg.get("driving_image"):
            job["drivers"] = [cfg["driving_image"]]
            if cfg.get("output_image"):
                job["output_files"] = [cfg["output_image"]]      # the reference's single (source, driver) -> one file contract
    if args.drivers_dir:
        exts = (".png", ".jpg", ".jpeg", ".bmp")
        job["drivers"] += [os.path.join(args.drivers_dir, f) for f in sorted(os.listdir(args.drivers_dir)) if f.lower().endswith(exts)]
    if not (job["source"] or job["source_tensor"]):
        raise SystemExit("reenact: no source (--source / --source-tensor / --config)")
    if not (job["drivers"] or job["drivers_tensor"]):
        raise SystemExit("reenact: no driver frames (--drivers / --drivers-dir / --drivers-tensor / --config)")
    if not (job["checkpoint"] or args.random_init):
        raise SystemExit("reenact: --checkpoint is required (or --random-init for plumbing tests)")
    return job


def _load_tensor(path):
    import numpy as np
    import torch

    t = torch.from_numpy(np.load(path)) if path.endswith(".npy") else torch.load(path, map_location="cpu")
    return t.float()


def _load_image(path):
    """inference.py:10-19: RGB, ToTensor (HWC uint8 -> CHW float /255), Normalize(0.5, 0.5)."""
    import numpy as np
    import torch
    from PIL import Image

    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return ((torch.from_numpy(a).permute(2, 0, 1) - 0.5) / 0.5).unsqueeze(0)


def _save_image(path, frame, unit_range: bool):
    """inference.py:38-44 (its BGR<->RGB swap + cv2.imwrite nets out to an RGB file)."""
    import numpy as np
    from PIL import Image

    x = frame.detach().float().cpu().numpy().transpose(1, 2, 0)
    x = x if unit_range else (x + 1.0) / 2.0
    Image.fromarray((np.clip(x, 0.0, 1.0) * 255).astype(np.uint8)).save(path)


def launch_ranks(args, argv) -> int:
    import subprocess

    from . import dp  # noqa: F401  (import check before spawning)

    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "megaportrait_hack_amd.reenact"] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # where the import alias package lives
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    return subprocess.run(cmd, env=env).returncode


def shard_plan(job: dict, n_drivers: int, rank: int, world: int):
    """What rank `rank` of `world` does (`--gpus N`: drivers sharded by frame, contiguous ranges, no collective): (begin, end, the files
    it writes).  Pure host logic — tests/test_dp.py checks that the ranks' ranges and outputs partition the job exactly."""
    from . import dp

    b, e = dp.shard_range(n_drivers, rank, world)
    if job["output_tensor"]:
        return b, e, [job["output_tensor"] if world == 1 else f"{job['output_tensor']}.rank{rank}"]
    return b, e, [job["output_files"][i] if job["output_files"] else os.path.join(job["output_dir"], f"frame_{i:05d}.png") for i in range(b, e)]


def run(job: dict, args, rank: int, world: int) -> List[str]:
    import torch

    from . import _lib, checkpoint, dp, gbase

    _lib.load()                                            # fail loudly if the HIP extension is missing
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    g = gbase.Gbase()
    if job["checkpoint"]:
        missing, unexpected = checkpoint.load_gbase(g, job["checkpoint"], strict=False)   # inference.py:60
        if missing or unexpected:
            print(f"reenact: checkpoint loaded with {len(missing)} missing / {len(unexpected)} unexpected keys", file=sys.stderr)
    g = g.to(dev).eval()
    if args.channels_last:
        torch.backends.cudnn.benchmark = True   # MIOpen find mode: the NHWC kernels only pay off with it (gbase.channels_last_2d)
        g.channels_last_2d()
    xs = _load_tensor(job["source_tensor"]) if job["source_tensor"] else _load_image(job["source"])
    n = _load_tensor(job["drivers_tensor"]).shape[0] if job["drivers_tensor"] else len(job["drivers"])
    b, e, outputs = shard_plan(job, n, rank, world)
    if job["drivers_tensor"]:
        xd = _load_tensor(job["drivers_tensor"])[b:e]
    else:
        xd = torch.cat([_load_image(p) for p in job["drivers"][b:e]], dim=0) if e > b else xs[:0]
    if not args.any_size and tuple(xs.shape[2:]) != (512, 512):
        raise SystemExit(f"reenact: the reference's Gbase only runs 512x512 frames (model.py:1157); got {tuple(xs.shape[2:])} "
                         "(pass --any-size to run other sizes)")
    frames = g.reenact(xs.to(dev), xd.to(dev), chunk=args.chunk, fp16=args.fp16)   # this rank's shard; no collective
    written = []
    if job["output_tensor"]:
        torch.save({"begin": b, "end": e, "frames": frames.cpu()}, outputs[0])
        written.append(outputs[0])
    else:
        os.makedirs(job["output_dir"], exist_ok=True)
        for i, path in zip(range(b, e), outputs):
            _save_image(path, frames[i - b], args.unit_range)
            written.append(path)
    return written


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    job = resolve(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        print(json.dumps({"job": job, "gpus": args.gpus, "self_launch": "WORLD_SIZE" not in os.environ and args.gpus > 1}))
        return 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    if world != args.gpus:
        raise SystemExit(f"reenact: --gpus {args.gpus} but WORLD_SIZE={world}")
    written = run(job, args, int(os.environ.get("RANK", "0")), world)
    print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "frames": len(written), "first": written[:1]}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
