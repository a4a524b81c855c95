"""Frame-level data parallelism for the hot slice: one process per GPU, frames sharded across ranks.

The hot slice has no cross-frame operation (GroupNorm is per sample, model.py:1140-1180), so
inference needs NO data-path collective: each rank runs its contiguous slice of the batch and
keeps its outputs.  `all_gather_frames` exists for callers that want the full
result on every rank (and for the world_size-2 gloo tests); it is not used by bench.py.
Training-time gradient averaging over RCCL lives in training.py (allreduce_gradients / train_step).
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `n_frames` for `rank` (first n%world ranks get +1)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_frames, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_inputs(inputs: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    n = next(iter(inputs.values())).shape[0]
    if any(v.shape[0] != n for v in inputs.values()):
        raise ValueError("all inputs must share the frame (batch) dimension")
    b, e = shard_range(n, rank, world)
    return {k: v[b:e] for k, v in inputs.items()}


def run_sharded(fn: Callable[..., torch.Tensor], inputs: Dict[str, torch.Tensor], rank: int, world: int) -> torch.Tensor:
    """Runs `fn(**local_inputs)` on this rank's frames only (may be an empty slice -> empty result)."""
    local = shard_inputs(inputs, rank, world)
    if next(iter(local.values())).shape[0] == 0:
        return None
    return fn(**local)


def all_gather_frames(local: torch.Tensor, n_frames: int, frame_shape, group=None, device=None,
                      dtype=torch.float32) -> torch.Tensor:
    """Ragged all-gather of per-rank frame slices back into frame order (torch.distributed; backend
    nccl == RCCL on ROCm, gloo in the CPU tests).  Pads to the largest shard, gathers, trims.
    A rank whose shard is empty passes local=None: its pad buffer must still live where the collective runs — pass
    `device` (and `dtype`) there; by default the current CUDA device under nccl, the CPU under gloo."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    cap = max(counts)
    if local is not None:
        dev, dtype = local.device, local.dtype
    elif device is not None:
        dev = torch.device(device)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros((cap, *frame_shape), dtype=dtype, device=dev)
    if counts[rank]:
        buf[:counts[rank]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def cross_reenact(hot, vs, es, Rs, ts, zs, Rd, td, zd, rank: int = 0, world: int = 1, chunk: int = 16):
    """BASELINE config 5: ONE source x many driver frames, sharded by driver frame.
    Source-side work (S2C field, warp #1, G3d; model.py:1151-1160) depends only on the source, so it
    is computed once per rank; only the C2D field and the fused warp+depth-sum run per driver
    (model.py:1163-1171).  Results equal calling the hot slice per (source, driver) pair."""
    from . import ops

    if vs.shape[0] != 1:
        raise ValueError("cross_reenact expects a single source frame")
    b, e = shard_range(Rd.shape[0], rank, world)
    with torch.no_grad():
        w_s2c = hot.warp_generator_s2c(Rs, ts, zs, es)
        vc2d = hot.G3d(ops.warp_volume(vs, w_s2c))
        outs = []
        for i in range(b, e, chunk):
            j = min(e, i + chunk)
            n = j - i
            w_c2d = hot.warp_generator_c2d(Rd[i:j], td[i:j], zd[i:j], es.expand(n, -1).contiguous())
            outs.append(ops.warp_volume_dsum(vc2d, w_c2d))   # one shared source volume, n driver fields (no expanded copy)
    return torch.cat(outs, dim=0) if outs else None
