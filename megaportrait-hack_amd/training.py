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

import os
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


class OverlappedGradReducer:
    """Gradient averaging overlapped with backward, in place.

    Every bucket owns ONE pre-allocated flat fp32 buffer and each parameter's `.grad` is a view into it, so there is no
    `torch.cat` into a staging buffer and no copy back (allreduce_gradients does both: 2 x 264 MB of extra traffic per
    step for the hot slice).  A post-accumulate-grad hook on every parameter counts its bucket down; a complete bucket
    is all-reduced asynchronously right away — while backward is still producing the gradients of earlier layers
    (buckets follow reverse registration order, the order backward produces them: G2d first, then C2D / G3d / S2C,
    Eapp / Emtn last).  Buckets are launched strictly in index order, so every rank issues the same collectives in the
    same order even when its graph differs (an empty frame shard, parameters unused in forward such as
    adaptive_matrix_beta, model.py:958-963): whatever is still pending goes out in `finish()`.

        reducer = OverlappedGradReducer(model.parameters())          # after the process group exists
        reducer.prepare(); loss_fn(model, **shard).backward(); reducer.finish(); optimizer.step()

    Two contracts that differ from the reference's single-GPU loop, both guarded:
    * ONE backward per prepare()/finish() pair.  A second backward() without prepare() (gradient accumulation) would find
      every bucket already launched and its gradients would never be all-reduced: `_on_grad` raises instead.  Accumulate by
      calling prepare(zero=False) between micro-batches of a step that should NOT be reduced yet, or sum the losses first.
    * Every parameter's `.grad` is a view into a zero-filled bucket, so a parameter that took no part in the step (the
      reference's unused `adaptive_matrix_beta`, or everything on an empty-shard rank) holds a ZERO gradient, not None — an
      optimizer with weight decay or Adam-style step counters would still update it where the reference's loop skips it.
      `finish(skip_unused=True)` restores the reference's semantics: parameters that received no gradient on ANY rank get
      `.grad = None` for the optimizer step (one extra 1-byte-per-parameter all-reduce); prepare() re-attaches the views.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20, average: bool = True):
        import torch.distributed as dist

        self.group, self.average = group, average
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.buckets = gradient_buckets(params, bucket_bytes)
        self.flats, self.views, self.bucket_of = [], {}, {}
        for i, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=bucket[0].dtype, device=bucket[0].device)
            off = 0
            for p in bucket:
                view = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.views[p] = view
                self.bucket_of[p] = i
                p.grad = view
            self.flats.append(flat)
        self.launched_during_backward = 0
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b]
        self.prepare()

    def prepare(self, zero: bool = True) -> None:
        """Start of a step: zero the flat buffers in place (the `.grad` views stay) and reset the bookkeeping."""
        if zero:
            for flat in self.flats:
                flat.zero_()
        self._touched = set()
        for p, view in self.views.items():
            if p.grad is not view:          # someone called zero_grad(set_to_none=True): re-attach the view
                p.grad = view
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._works = []
        self._in_backward = True
        self.launched_during_backward = 0

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if self._next > self.bucket_of[p] or not self._in_backward:
            raise RuntimeError("OverlappedGradReducer: a gradient arrived for a bucket that was already all-reduced in this step "
                               "(second backward() without prepare(), or backward() after finish()); call prepare() first")
        self._touched.add(p)
        view = self.views[p]
        if p.grad is not view:               # autograd installed a fresh tensor (grad was None): fold it into the bucket
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
        i = self.bucket_of[p]
        self._pending[i] -= 1
        self._launch_ready()

    def _launch_ready(self, force: bool = False) -> None:
        import torch.distributed as dist

        while self._next < len(self.buckets) and (force or self._pending[self._next] <= 0):
            if self.world > 1:
                self._works.append(dist.all_reduce(self.flats[self._next], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if self._in_backward and not force:
                    self.launched_during_backward += 1
            self._next += 1

    def finish(self, skip_unused: bool = False) -> int:
        """After backward: launch what is still pending (buckets holding parameters without a gradient on this rank),
        wait for every collective, average in place.  Returns the number of all-reduces of this step.
        skip_unused: parameters no rank produced a gradient for get `.grad = None` (the optimizer then skips them, like the
        reference's loop does for `adaptive_matrix_beta`)."""
        import torch.distributed as dist

        self._in_backward = False
        self._launch_ready(force=True)
        for w in self._works:
            w.wait()
        if self.average and self.world > 1:
            for flat in self.flats:
                flat.div_(self.world)
        if skip_unused:
            order = [p for b in self.buckets for p in b]
            used = torch.tensor([p in self._touched for p in order], dtype=torch.uint8, device=self.flats[0].device if self.flats else "cpu")
            if self.world > 1:
                dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)
            for p, u in zip(order, used.tolist()):
                if not u:
                    p.grad = None
        return len(self._works)

    def remove(self) -> None:
        for h in self._handles:
            h.remove()


def train_step(model: torch.nn.Module, loss_fn: Callable[..., torch.Tensor], optimizer: torch.optim.Optimizer,
               inputs: Dict[str, torch.Tensor], group=None, reducer: Optional[OverlappedGradReducer] = None,
               pack_table=None) -> torch.Tensor:
    """zero_grad -> forward -> loss -> backward -> (gradient all-reduce when distributed) -> optimizer.step().
    `loss_fn(model, **inputs)` returns the scalar loss of this rank's shard.  With `reducer` the all-reduces run
    bucket by bucket underneath backward on pre-allocated flat gradient buffers; without, after backward
    (allreduce_gradients).  `pack_table` (ops.PackTable.from_module(model), built after one step has run): every conv weight
    the step uses is re-packed in <= 5 launches at its top instead of two or three launches per weight at first use — the
    distributed (eager) step's counterpart of what GraphedTrainStep captures; same bits."""
    import torch.distributed as dist

    if pack_table is not None:
        pack_table.run()
    if reducer is not None:
        reducer.prepare()
        loss = loss_fn(model, **inputs)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return loss.detach()
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
    re-packing of the updated weights — as one batched table run at the top of the step (ops.PackTable, `batched_packs`).  Everything on the path zero-fills with kernels, not hipMemsetAsync: memset
    nodes of a captured graph were not reliably ordered with the kernels around them on ROCm 7.2 (a replayed step
    went wrong in ~40 % of runs — stale f16x3 pack headers — until they were replaced; csrc/mphip_common.h).  Single-process only: a distributed step keeps the eager `train_step`
    (collectives stay outside the graph).

    The stale loss of r03-r04, root-caused in r05: the step was always right, the LOSS TENSOR was not
    rewritten.  `F.mse_loss` / `.mean()` over more than ~64 k elements is ATen's multi-block reduction: it zeroes a semaphore array with
    hipMemsetAsync (a MEMSET node in the captured graph) and the last block to finish writes the result.  On ROCm 7.x a memset node is
    not reliably ordered with the kernel nodes around it — even in this strictly linear chain (262 nodes, 261 edges, one memset:
    tools/dbg_graph_topology.py) — so the reduction sometimes starts on the previous replay's count, no block is "the last one" and the
    output keeps its old value (5-9 stale losses in 12 replays with the batched re-pack at the top of the graph, ~3 in 20 before it;
    a loss reduced in two single-block stages was never stale; tools/dbg_replay_stale_loss.py).  Not reproducible with torch kernels
    alone (tools/repro_graph_memset.py).  Fix: after capture every MEMSET node of the graph is rewritten as a kernel node
    (mphip_graph_memsets_to_kernels, the same cure as for the library's own memsets above); `memset_nodes_replaced` counts them.
    `__call__` returns a private, stream-ordered copy of the loss (one small copy kernel on the replay's stream, no host wait), so replays
    pipeline like eager steps and losses kept across steps do not alias.  `sync_after_replay=True` adds a device-wide wait after every
    replay — r05's belt-and-braces default from before the root cause was known; it is the default only when the memset rewrite is
    switched off (MPHIP_GRAPH_MEMSET_FIX=0)."""

    def __init__(self, model: torch.nn.Module, loss_fn: Callable[..., torch.Tensor], optimizer: torch.optim.Optimizer,
                 example_inputs: Dict[str, torch.Tensor], warmup: int = 3, sync_after_replay: Optional[bool] = None,
                 batched_packs: Optional[bool] = None):
        import copy

        from . import ops

        for group in optimizer.param_groups:   # fail here with a clear message, not inside the capture
            if "capturable" in group and not group["capturable"]:
                raise ValueError(f"GraphedTrainStep: {type(optimizer).__name__} must be built with capturable=True to be "
                                 "captured in a hipGraph (its step counters live on the host otherwise); SGD works as is")
        self.model, self.optimizer = model, optimizer
        if sync_after_replay is None:
            sync_after_replay = os.environ.get("MPHIP_GRAPH_MEMSET_FIX", "1") == "0"
        self.sync_after_replay = bool(sync_after_replay)
        self.static_in = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in example_inputs.items()}
        # the warm-up runs REAL steps (the allocator and the packed-weight caches must see the final shapes): parameters,
        # buffers and optimizer state are snapshotted and restored, so building the graph does not train the model
        model_state = copy.deepcopy(model.state_dict())
        had_state = len(optimizer.state) > 0
        optim_state = copy.deepcopy(optimizer.state_dict()) if had_state else None
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
        with torch.no_grad():
            model.load_state_dict(model_state)      # copies in place: parameter storage (captured below) is unchanged
            if had_state:
                optimizer.load_state_dict(optim_state)
            else:
                # a fresh optimizer: keep the state tensors the warm-up allocated (the capture below must record the steady-state
                # update, not the first-step "create the buffer" branch) but reset their VALUES — zero momentum / moments / step
                # counts are what a fresh SGD(momentum, dampening=0) or Adam(capturable=True) starts from
                for st in optimizer.state.values():
                    for v in st.values():
                        if isinstance(v, torch.Tensor):
                            v.zero_()
        ops.invalidate_packs()
        # keep_graph: the captured hipGraph_t stays reachable (raw_cuda_graph()) and is instantiated by hand below, after its MEMSET
        # nodes were rewritten as kernel nodes (see the class docstring; MPHIP_GRAPH_MEMSET_FIX=0: dev A/B, leaves them)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        optimizer.zero_grad(set_to_none=True)
        for v in self.static_in.values():
            v.grad = None
        ops.begin_capture()   # range descriptors measured on the warm-up batch are not frozen into the graph (ops._range_for)
        # batched_packs: the re-packing of every weight the warm-up steps packed (forward and bwd-data direction, the precisions they
        # used) is captured as ONE table run at the top of the step (<= 5 launches, same bits) instead of two or three launches per
        # weight at its first use: r05's step spent 1.08 of 10.9 ms in 105 latency-bound pack launches
        if batched_packs is None:
            batched_packs = os.environ.get("MPHIP_BATCHED_PACKS", "1") != "0"   # (dev: same-box A/B)
        self.pack_table = ops.PackTable.from_module(model) if batched_packs else None
        with ops.repack_always(), torch.cuda.graph(self.graph):
            if self.pack_table is not None:
                self.pack_table.run()
            self.static_loss = loss_fn(model, **self.static_in)
            self.static_loss.backward()
            optimizer.step()
        self.memset_nodes_replaced = ops.finish_graph_capture(self.graph)

    def __call__(self, **inputs) -> torch.Tensor:
        from . import ops

        with torch.no_grad():
            for k, v in inputs.items():
                if v.data_ptr() != self.static_in[k].data_ptr():
                    self.static_in[k].copy_(v)
        self.graph.replay()
        ops.invalidate_packs()  # the replay rewrote the parameters without touching their version counters
        if self.sync_after_replay:
            torch.cuda.synchronize()
        return self.static_loss.detach().clone()   # stream-ordered after the replay; never the static tensor itself
