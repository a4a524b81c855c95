"""Multi-process tests of the frame-sharding path on CPU (gloo, world_size 2): shard planner,
ragged gather order, and that sharded execution reproduces the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from megaportrait_hack_amd import dp


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 1024):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(4, 2, 2)


def _fake_hot(vs, scale):
    # per-frame function with no cross-frame coupling, like the hot slice
    return vs.flatten(1).sum(dim=1, keepdim=True) * scale


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        inputs = {"vs": torch.randn(n_frames, 3, 4, generator=g), "scale": torch.arange(n_frames, dtype=torch.float32).view(-1, 1) + 1}
        local = dp.run_sharded(_fake_hot, inputs, rank, world)
        full = dp.all_gather_frames(local, n_frames, (1,))
        want = _fake_hot(**inputs)
        ok = torch.equal(full, want)
        b, e = dp.shard_range(n_frames, rank, world)
        ok = ok and ((local is None and b == e) or torch.equal(local, want[b:e]))
        # weak-scaling timing reduction used by bench.py: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == world
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 5, 1])
def test_sharded_equals_single_process_gloo(n_frames):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
