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


# ------------------------------------------------------------------ training: bucketed gradient averaging (row f2 x e)
def test_gradient_buckets_cover_every_trainable_parameter_once():
    from megaportrait_hack_amd import training

    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 3000, 7, 500, 1)]
    ps[2].requires_grad_(False)
    buckets = training.gradient_buckets(ps, bucket_bytes=4 * 2000)
    flat = [p for b in buckets for p in b]
    assert [id(p) for p in flat] == [id(p) for p in reversed([ps[0], ps[1], ps[3], ps[4]])]   # backward order
    assert all(sum(p.numel() * 4 for p in b) <= 4 * 2000 or len(b) == 1 for b in buckets)      # an oversized tensor rides alone


def _train_worker(rank, world, port, q):
    """Two ranks, each with its shard of a batch: after allreduce_gradients + SGD every rank holds the parameters a
    single process reaches on the whole batch (loss = mean over frames, shards equal in size)."""
    from megaportrait_hack_amd import training

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(11)
        def make():
            m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
            m.unused = torch.nn.Parameter(torch.ones(3))   # never used in forward: grad stays None on every rank (adaptive_matrix_beta)
            return m
        ref, model = make(), make()
        model.load_state_dict(ref.state_dict())
        g = torch.Generator().manual_seed(3)
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
        loss_fn = lambda m, x, y: torch.nn.functional.mse_loss(m(x), y)
        opt_ref = torch.optim.SGD(ref.parameters(), lr=0.1)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        for _ in range(3):
            training.train_step(model, loss_fn, opt, dp.shard_inputs({"x": x, "y": y}, rank, world))
            opt_ref.zero_grad(set_to_none=True)
            loss_fn(ref, x, y).backward()
            opt_ref.step()
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(model.parameters(), ref.parameters()))
        n_calls = training.allreduce_gradients(model.parameters(), bucket_bytes=64)   # tiny buckets: several collectives
        ok = ok and n_calls >= 2
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_train_step_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _overlap_worker(rank, world, port, q):
    """The hook path (OverlappedGradReducer: flat in-place buckets, all-reduce launched from post-accumulate-grad hooks
    while backward is still running) must walk the parameters exactly like the post-hoc path (allreduce_gradients),
    also when one rank's shard is EMPTY (its loss is a zero connected to the parameters) and with a parameter that
    never gets a gradient."""
    from megaportrait_hack_amd import training

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(11)

        def make():
            m = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
            m.unused = torch.nn.Parameter(torch.ones(3))
            return m

        posthoc, hooked = make(), make()
        hooked.load_state_dict(posthoc.state_dict())
        g = torch.Generator().manual_seed(3)
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)

        def loss_fn(m, x, y):
            if x.shape[0] == 0:      # empty shard: zero loss that still reaches every used parameter
                return sum(p.sum() for n, p in m.named_parameters() if n != "unused") * 0.0
            return torch.nn.functional.mse_loss(m(x), y, reduction="sum") / 8.0

        opt_p = torch.optim.SGD(posthoc.parameters(), lr=0.1, momentum=0.5)
        opt_h = torch.optim.SGD(hooked.parameters(), lr=0.1, momentum=0.5)
        reducer = training.OverlappedGradReducer(hooked.parameters(), bucket_bytes=600, average=False)   # several small buckets
        assert len(reducer.buckets) >= 3
        early = 0
        for step in range(4):
            if step < 2:
                shard = dp.shard_inputs({"x": x, "y": y}, rank, world)
            else:                    # ragged: rank 0 holds everything, rank 1 an empty shard
                shard = {"x": x, "y": y} if rank == 0 else {"x": x[:0], "y": y[:0]}
            training.train_step(hooked, loss_fn, opt_h, shard, reducer=reducer)
            early += reducer.launched_during_backward
            opt_p.zero_grad(set_to_none=True)
            loss_fn(posthoc, **shard).backward()
            training.allreduce_gradients(posthoc.parameters(), bucket_bytes=600, average=False)
            opt_p.step()
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(hooked.parameters(), posthoc.parameters()))
        ok = ok and early >= 4                        # buckets did go out from inside backward
        ok = ok and all(p.grad is reducer.views[p] for b in reducer.buckets for p in b)   # grads still live in the flat buffers
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_allreduce_equals_posthoc_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_overlapped_reducer_guards_single_process():
    """(a) a second backward() without prepare() must raise instead of silently skipping the all-reduce;
    (b) finish(skip_unused=True) gives parameters that took no part in the step `.grad = None` (the reference's single-GPU
    loop never updates `adaptive_matrix_beta`; an Adam / weight-decay step on a zero gradient would)."""
    from megaportrait_hack_amd import training

    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
    m.unused = torch.nn.Parameter(torch.ones(3))
    red = training.OverlappedGradReducer(m.parameters(), bucket_bytes=64)
    x = torch.randn(5, 4)
    red.prepare()
    m(x).sum().backward()
    with pytest.raises(RuntimeError, match="prepare"):
        m(x).sum().backward()          # gradient accumulation without prepare(): every bucket of the step already went out
    red.prepare()
    m(x).sum().backward()
    red.finish()
    assert m.unused.grad is not None and float(m.unused.grad.abs().max()) == 0.0   # default: zero gradient (documented)
    red.prepare()
    m(x).sum().backward()
    red.finish(skip_unused=True)
    assert m.unused.grad is None and all(p.grad is not None for p in m[0].parameters())
    opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=0.1)
    before = m.unused.detach().clone()
    opt.step()
    assert torch.equal(m.unused, before)                                            # skipped, like the reference's loop
    red.prepare()                                                                   # views are re-attached for the next step
    assert m.unused.grad is red.views[m.unused]


def test_empty_shard_loss_is_multiply_free():
    """the empty-shard connection to the parameters must not turn an Inf/NaN weight into a NaN loss."""
    ps = [torch.nn.Parameter(torch.tensor([1.0, float("inf")])), torch.nn.Parameter(torch.tensor([float("nan")]))]
    z = sum(p.reshape(-1)[:0].sum() for p in ps)
    assert float(z) == 0.0
    z.backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in ps)


@pytest.mark.parametrize("n_drivers,world", [(1024, 8), (10, 3), (3, 8), (1, 1), (0, 2)])
def test_reenact_cli_shards_partition_the_job(n_drivers, world, tmp_path):
    """`python -m megaportrait_hack_amd.reenact --gpus N` (BASELINE config 5: 1 source x 1024 drivers over 8 ranks): the ranks' driver
    ranges are contiguous, disjoint and cover every frame, ragged and empty shards included, and no two ranks write the same file —
    in both output modes (PNG per frame / one tensor file per rank)."""
    from megaportrait_hack_amd import reenact

    for tensor_mode in (False, True):
        job = {"output_tensor": str(tmp_path / "out.pt") if tensor_mode else None, "output_files": None, "output_dir": str(tmp_path / "o")}
        covered, files = [], []
        for rank in range(world):
            b, e, outs = reenact.shard_plan(job, n_drivers, rank, world)
            assert 0 <= b <= e <= n_drivers
            covered += list(range(b, e))
            files += outs
            assert tensor_mode or len(outs) == e - b
        assert covered == list(range(n_drivers))                 # in rank order: contiguous shards, nothing twice, nothing missing
        assert len(set(files)) == len(files)
    # the self-launcher carries --gpus through (dry run: no GPU, no process spawned)
    import json, subprocess, sys
    r = subprocess.run([sys.executable, "-m", "megaportrait_hack_amd.reenact", "--source-tensor", "s.pt", "--drivers-tensor", "d.pt",
                        "--output-tensor", "o.pt", "--random-init", "--gpus", str(max(world, 1)), "--dry-run"], capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-500:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["gpus"] == max(world, 1) and plan["self_launch"] == (world > 1)
