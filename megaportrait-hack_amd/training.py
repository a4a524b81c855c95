"""Training-side host code of the hot slice (scope row f2): data-parallel gradient averaging over RCCL and a
hipGraph-replayed training step.

The reference trains on one GPU with plain `loss.backward(); optimizer.step()` (train.py:194-330).  Here the same
step scales the way SURVEY.md §8e prescribes — one process per GPU, frames sharded across ranks (dp.shard_inputs),
gradients averaged with a few large bucketed all-reduces (backend "nccl" IS RCCL on ROCm; xGMI rings are per-link
bound, so buckets are large and few) — and the ~1000 small stream-ordered launches of a forward+backward+SGD step
are captured once into a hipGraph and replayed with one launch (the step is otherwise host-bound: FlowField's
layers are tiny).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch


def gradient_buckets(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 256 << 20) -> List[List[torch.nn.Parameter]]:
    """Parameters that have a gradient, in reverse registration order (the order backward produces them), packed into
    buckets of about `bucket_bytes`.  G3d's 250 MB of weights -> 1-2 all-reduces per step instead of one per tensor."""
    ps = [p for p in params if p.requires_grad]
    buckets, cur, size = [], [], 0
    for p in reversed(ps):
        n = p.numel() * p.element_size()
        if cur and size + n > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += n
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 256 << 20,
                        average: bool = True) -> int:
    """Averages .grad over the ranks of `group` with one flat all-reduce per bucket (async, then unflattened).
    Parameters whose grad is None on this rank (unused: adaptive_matrix_beta, model.py:958-963) contribute zeros so
    every rank issues the same collectives.  Returns the number of all-reduce calls."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return 0
    works = []
    for bucket in gradient_buckets(params, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
    for work, flat, bucket in works:
        work.wait()
        if average:
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len(works)


def train_step(model: torch.nn.Module, loss_fn: Callable[..., torch.Tensor], optimizer: torch.optim.Optimizer,
               inputs: Dict[str, torch.Tensor], group=None) -> torch.Tensor:
    """zero_grad -> forward -> loss -> backward -> (gradient all-reduce when distributed) -> optimizer.step().
    `loss_fn(model, **inputs)` returns the scalar loss of this rank's shard."""
    import torch.distributed as dist

    optimizer.zero_grad(set_to_none=True)
    loss = loss_fn(model, **inputs)
    loss.backward()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        allreduce_gradients(model.parameters(), group=group)
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """One training step (forward + backward + optimizer update) captured as a hipGraph for fixed input shapes.

    Inputs are copied into static buffers; `__call__` replays the graph and returns the (static) loss tensor.
    The packed-weight caches are bypassed while capturing (ops.repack_always) so the graph contains the per-step
    re-packing of the updated weights.  Everything on the path zero-fills with kernels, not hipMemsetAsync: memset
    nodes of a captured graph were not reliably ordered with the kernels around them on ROCm 7.2 (a replayed step
    went wrong in ~40 % of runs — stale f16x3 pack headers — until they were replaced; csrc/mphip_common.h).  Single-process only: a distributed step keeps the eager `train_step`
    (collectives stay outside the graph)."""

    def __init__(self, model: torch.nn.Module, loss_fn: Callable[..., torch.Tensor], optimizer: torch.optim.Optimizer,
                 example_inputs: Dict[str, torch.Tensor], warmup: int = 3):
        from . import ops

        self.model, self.optimizer = model, optimizer
        self.static_in = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in example_inputs.items()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                loss_fn(model, **self.static_in).backward()
                optimizer.step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        for v in self.static_in.values():
            v.grad = None
        with ops.repack_always(), torch.cuda.graph(self.graph):
            self.static_loss = loss_fn(model, **self.static_in)
            self.static_loss.backward()
            optimizer.step()

    def __call__(self, **inputs) -> torch.Tensor:
        from . import ops

        with torch.no_grad():
            for k, v in inputs.items():
                if v.data_ptr() != self.static_in[k].data_ptr():
                    self.static_in[k].copy_(v)
        self.graph.replay()
        ops.invalidate_packs()  # the replay rewrote the parameters without touching their version counters
        return self.static_loss
