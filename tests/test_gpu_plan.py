"""GPU tests of the one-call entries (include/mphip.h: mphip_hot_slice_plan_*, mphip_g3d_forward).
The plan issues the same launches as the Python schedule (model._HotSliceRunner._run), so the results must be BITWISE
equal to it — and through it pinned to the oracle / the reference's goldens (tests/test_gpu_parity.py)."""
import os
import subprocess

import pytest
import torch

from oracle import hotpath_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    from megaportrait_hack_amd import _lib, model

    _lib.load()
    return model


def _hot(M, dev, seed=21):
    hot = M.GbaseHotSlice()
    M.load_hot_state_dict(hot, R.seeded_gbase_hot_state_dict(seed))
    return hot.to(dev).eval()


@pytest.mark.parametrize("shape", [(1, 16, 16, 16), (3, 8, 16, 16), (2, 16, 64, 64)])
def test_plan_equals_python_schedule_bitwise(dev, M, shape):
    from megaportrait_hack_amd import plan as P

    b, d, h, w = shape
    hot = _hot(M, dev)
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(b, 5, D=d, H=h, W=w).items()}
    pl = P.HotSlicePlan(hot, dims=(96, d, h, w))
    with torch.no_grad():
        want = hot._run_python(check_shape=False, **inp)
        got = pl(**inp)
        again = pl(**inp)      # the workspace is reused: same result
    assert got.shape == want.shape == (b, 96, h, w)
    assert torch.equal(got, want)
    assert torch.equal(again, want)
    # ... and against the CPU oracle at the small sizes
    if d * h * w <= 16 * 16 * 16:
        ref = R.hot_slice(sd=R.seeded_gbase_hot_state_dict(21), **R.seeded_hot_inputs(b, 5, D=d, H=h, W=w))
        assert (got.cpu() - ref).abs().max().item() < 1e-3
    single = P.HotSlicePlan(hot, dims=(96, d, h, w), single_stream=True)
    with torch.no_grad():
        assert torch.equal(single(**inp), want)


def test_plan_is_the_default_inference_path(dev, M):
    """GbaseHotSlice.forward under no_grad goes through the plan (one ctypes call); MPHIP_C_PLAN=0 / use_c_plan=False
    selects the per-op Python schedule.  Same bits either way."""
    hot = _hot(M, dev)
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 7, D=8, H=16, W=16).items()}
    with torch.no_grad():
        a = hot.forward_any_size(**inp)
        assert hot.__dict__.get("_plans"), "the plan was not used"
        hot.use_c_plan = False
        b = hot.forward_any_size(**inp)
    assert torch.equal(a, b)


def test_one_plan_per_caller_stream(dev, M):
    """A plan owns one side stream and one fork/join event pair, so GbaseHotSlice keeps one plan per caller stream (two batches in
    flight on two streams = two independent plans: bench.py's default loop).  Same bits on every stream, interleaved calls included;
    the table is bounded (least recently used plans are closed)."""
    hot = _hot(M, dev)
    a = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 7, D=8, H=16, W=16).items()}
    b = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 8, D=8, H=16, W=16).items()}
    with torch.no_grad():
        want_a, want_b = hot.forward_any_size(**a).clone(), hot.forward_any_size(**b).clone()
        assert len(hot.__dict__["_plans"]) == 1
        lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for ln in lanes:
            ln.wait_stream(torch.cuda.current_stream())
        outs = []
        for i in range(8):   # interleaved, nothing waits on anything in between
            with torch.cuda.stream(lanes[i % 2]):
                outs.append(hot.forward_any_size(**(a if i % 2 == 0 else b)))
        torch.cuda.synchronize()
        assert len(hot.__dict__["_plans"]) == 3
        for i, o in enumerate(outs):
            assert torch.equal(o, want_a if i % 2 == 0 else want_b)
        many = [torch.cuda.Stream(device=dev) for _ in range(hot._MAX_PLANS + 2)]
        for st in many:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                assert torch.equal(hot.forward_any_size(**a), want_a)
        assert len(hot.__dict__["_plans"]) == hot._MAX_PLANS
        assert torch.equal(hot.forward_any_size(**b), want_b)   # (the default stream's plan was evicted and is rebuilt)


def test_plan_honours_the_conv_precision_switch(dev, M):
    """ops.set_conv_precision('fp32') (bench.py's `fp32_exact` leg, MPHIP_CONV_PRECISION) must reach the plan's launches."""
    from megaportrait_hack_amd import ops

    hot = _hot(M, dev)
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 11, D=16, H=32, W=32).items()}
    old = ops.get_conv_precision()
    try:
        with torch.no_grad():
            fast = hot.forward_any_size(**inp)
            ops.set_conv_precision("fp32")
            exact = hot.forward_any_size(**inp)
            want = hot._run_python(check_shape=False, **inp)
            ops.set_conv_precision("auto")
            fast2 = hot.forward_any_size(**inp)
    finally:
        ops.set_conv_precision(old)
    assert torch.equal(exact, want) and torch.equal(fast, fast2)
    assert not torch.equal(exact, fast)          # different arithmetic (f16x3 vs exact fp32), same answer to fp32 class
    assert (exact - fast).abs().max().item() < 1e-4


def test_plan_follows_weight_updates(dev, M):
    from megaportrait_hack_amd import plan as P

    hot = _hot(M, dev)
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(1, 9, D=8, H=16, W=16).items()}
    pl = P.HotSlicePlan(hot, dims=(96, 8, 16, 16))
    with torch.no_grad():
        before = pl(**inp).clone()
        hot.G3d.final_conv.weight.mul_(1.5)                              # in-place update: version counter bumps
        hot.warp_generator_c2d.flowfield.conv1x1.bias.add_(0.25)
        after = pl(**inp)
        want = hot._run_python(check_shape=False, **inp)
    assert not torch.equal(before, after)
    assert torch.equal(after, want)
    M.load_hot_state_dict(hot, R.seeded_gbase_hot_state_dict(33))        # load_state_dict copies in place
    with torch.no_grad():
        assert torch.equal(pl(**inp), hot._run_python(check_shape=False, **inp))


def test_g3d_forward_entry(dev, M):
    from megaportrait_hack_amd import ops, plan as P

    hot = _hot(M, dev)
    x = R.seeded_tensor((2, 96, 8, 16, 16), 41).to(dev)
    pl = P.HotSlicePlan(hot, dims=(96, 8, 16, 16), g3d_only=True)
    with torch.no_grad():
        want = hot.G3d(x.clone())
        got = pl.g3d(x)                                  # x_range = NULL: the library measures x
        rng = ops.absmax_range(x)
        got2 = pl.g3d(x, rng)
    assert torch.equal(got, want) and torch.equal(got2, want)
    with pytest.raises(RuntimeError, match="generators"):
        pl(**{k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 5, D=8, H=16, W=16).items()})


def test_plan_argument_errors(dev, M):
    from megaportrait_hack_amd import plan as P

    hot = _hot(M, dev)
    pl = P.HotSlicePlan(hot, dims=(96, 8, 16, 16))
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(1, 5, D=8, H=16, W=16).items()}
    with pytest.raises(RuntimeError, match="does not match"):
        pl(**{**inp, "vs": inp["vs"][:, :, :4].contiguous()})
    lib = pl.lib
    need = lib.mphip_hot_slice_workspace_bytes(pl._handle, 1)
    assert need > 0
    small = torch.empty(1024, dtype=torch.uint8, device=dev)
    import ctypes
    out = torch.empty((1, 96, 16, 16), device=dev)
    rc = lib.mphip_hot_slice_forward(pl._handle, *(ctypes.c_void_p(inp[k].data_ptr()) for k in ("vs", "es", "Rs", "ts", "zs", "Rd", "td", "zd")),
                                     ctypes.c_void_p(out.data_ptr()), 1, ctypes.c_void_p(small.data_ptr()), small.numel(), None)
    assert rc == -3 and b"workspace" in lib.mphip_last_error()


def test_plan_graph_capture(dev, M):
    """No allocation / host sync in forward: the one-call entry is capturable after a warm-up call."""
    from megaportrait_hack_amd import plan as P

    hot = _hot(M, dev)
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(1, 5, D=8, H=16, W=16).items()}
    pl = P.HotSlicePlan(hot, dims=(96, 8, 16, 16))
    with torch.no_grad():
        want = pl(**inp).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pl(**inp)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = pl(**inp)
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_graphed_hot_slice_goes_through_the_warmed_plan(dev, M):
    """model.GraphedHotSlice with the default C-plan path: it warms up on a side stream and captures under
    torch.cuda.graph(), which switches to a separate capture stream — the plan that was warmed up must be the one that is captured (a
    plan built during the capture would hipMalloc / copy synchronously / pack weights inside it), replays must follow new inputs and
    equal the eager forward bit for bit, and asking for an unseen shape while capturing must fail loudly."""
    hot = _hot(M, dev)
    assert hot.use_c_plan
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 5, D=8, H=16, W=16).items()}
    inp2 = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 9, D=8, H=16, W=16).items()}
    with torch.no_grad():
        want, want2 = hot.forward_any_size(**inp).clone(), hot.forward_any_size(**inp2).clone()
    plans_before = len(hot.__dict__.get("_plans", {}))
    graphed = M.GraphedHotSlice(hot, inp, any_size=True)
    assert len(hot.__dict__["_plans"]) == plans_before + 1          # the warm-up stream's plan; none created under capture
    assert torch.equal(graphed(**inp), want)
    assert torch.equal(graphed(**inp2), want2)
    assert torch.equal(graphed(**inp), want)
    other = {k: v.to(dev) for k, v in R.seeded_hot_inputs(2, 5, D=8, H=16, W=32).items()}
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="no warmed-up plan"):
        with torch.no_grad(), torch.cuda.graph(g):
            hot.forward_any_size(**other)
    torch.cuda.synchronize()


def test_plan_from_plain_c(plan_c_exe):
    """tests/c_abi/plan_smoke.c: K2 coordinates / indices / values bit-exact vs the C oracle, K3, an f16x3 3x3x3 conv with a
    caller-built range descriptor, and the whole slice through mphip_hot_slice_plan_* against the committed oracle golden
    (tests/golden/plan_c_expected.bin, oracle/make_golden_plan.py) — no Python between the C program and the library."""
    gold = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([plan_c_exe, os.path.join(gold, "plan_c_manifest.txt"), os.path.join(gold, "plan_c_expected.bin")],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-exact" in r.stdout and "PLAN C ABI OK" in r.stdout


def _rand_boxes(n, d, h, w, gen):
    lo = torch.stack([torch.randint(0, w - 2, (n,), generator=gen), torch.randint(0, h - 2, (n,), generator=gen),
                      torch.randint(0, d - 1, (n,), generator=gen)], dim=1)
    ext = torch.stack([torch.randint(1, 7, (n,), generator=gen), torch.randint(1, 7, (n,), generator=gen), torch.randint(1, 4, (n,), generator=gen)], dim=1)
    ext = torch.minimum(ext, torch.tensor([w, h, d]) - lo)
    return torch.cat([lo, ext, torch.zeros(n, 2, dtype=torch.long)], dim=1).int()


@pytest.mark.parametrize("shape", [(2, 96, 8, 16, 32), (3, 96, 4, 16, 16), (1, 96, 16, 64, 64)])
def test_demand_driven_conv_and_upsample_equal_the_full_ops_inside_the_boxes(dev, shape):
    """mphip_conv3d_fwd_roi / mphip_upsample_trilinear2_roi write exactly the full ops' bits wherever a box's tiles (and their
    halos) reach, and NOTHING else (the output starts as NaN and must stay NaN outside)."""
    import ctypes

    from megaportrait_hack_amd import _lib, ops

    lib = _lib.load()
    P = ctypes.c_void_p
    n, c, d, h, w = shape
    gen = torch.Generator().manual_seed(5)
    x_lo = torch.randn(n, c, d // 2, h // 2, w // 2, generator=gen).to(dev)
    pc = ops.PackedConv((torch.randn(96, c, 3, 3, 3, generator=gen) * 0.03).to(dev), torch.randn(96, generator=gen).to(dev))
    boxes = _rand_boxes(n, d, h, w, gen).to(dev)
    tile = (ctypes.c_int * 3)()
    assert lib.mphip_conv3d_roi_granule(n, c, 96, d, h, w, 3, 1, tile) == 1
    td, th, tw = tile[0], tile[1], tile[2]
    st = P(torch.cuda.current_stream().cuda_stream)
    up_full = ops.upsample_trilinear2(x_lo)
    up_roi = torch.full_like(up_full, float("nan"))
    _lib.check(lib.mphip_upsample_trilinear2_roi(P(x_lo.data_ptr()), P(up_roi.data_ptr()), P(boxes.data_ptr()), 0, n, c, d // 2, h // 2, w // 2,
                                                 td, th, tw, st), "upsample roi")
    y_full = ops.conv3d(up_full, pc, precision=1)
    rng = ops.tensor_range(up_full)
    # the demand-driven conv reads the demand-driven upsample: NaN anywhere it should not look would surface in y
    y_roi = torch.full_like(y_full, float("nan"))
    ws_bytes = lib.mphip_conv3d_roi_workspace_bytes(n, c, 96, d, h, w, 3, 1)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    _lib.check(lib.mphip_conv3d_fwd_roi(P(up_roi.data_ptr()), P(rng.data_ptr()), P(pc.packed(1).data_ptr()), P(pc.bias.data_ptr()), P(y_roi.data_ptr()),
                                        P(boxes.data_ptr()), 0, n, c, 96, d, h, w, 3, 1, P(ws.data_ptr()), ws.numel(), st), "conv roi")
    need = torch.zeros((n, d, h, w), dtype=torch.bool)
    for i, (lx, ly, lz, ex, ey, ez, _, _) in enumerate(boxes.cpu().tolist()):
        need[i, lz // td * td:((lz + ez - 1) // td + 1) * td, ly // th * th:((ly + ey - 1) // th + 1) * th, lx // tw * tw:((lx + ex - 1) // tw + 1) * tw] = True
    need = need.to(dev)[:, None].expand_as(y_full)
    assert torch.equal(y_roi[need], y_full[need])             # bit-identical where asked for
    if lib.mphip_conv3d_splits(n, c, 96, d, h, w, 3, 1) == 1:   # (a split-K launch reduces its slabs over the whole tensor)
        assert torch.isnan(y_roi[~need]).all()                # untouched elsewhere
    assert need.float().mean().item() < 1.0 or shape[2] <= 4


def test_demand_driven_conv_on_a_travelling_field_costs_what_the_full_launch_costs(dev):
    """when the final warp's samples travel through the whole volume the boxes are the volume and the demand-driven
    launch lists every tile — it must then cost what the plain launch costs (r03: three workgroups per listed tile re-staged X three
    times).  Since r04 both are the SAME kernel (the F(2,3) kernel walks either the full range or the list): within 8 % of each other
    on the graded shape (the list launch adds the 5 us tile-list kernel and loses the XCD-aware tile order), same bits."""
    from megaportrait_hack_amd import ops

    n, c, d, h, w = 8, 96, 16, 64, 64
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, d, h, w, generator=gen).to(dev)
    pc = ops.PackedConv((torch.randn(96, c, 3, 3, 3, generator=gen) * 0.03).to(dev), torch.randn(96, generator=gen).to(dev))
    boxes = torch.tensor([[0, 0, 0, w, h, d, 0, 0]] * n, dtype=torch.int32).to(dev)
    rng = ops.absmax_range(x)

    def timed(fn):
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10, y

    t_full, y_full = timed(lambda: ops.conv3d(x, pc, precision=1, x_range=rng))
    t_roi, y_roi = timed(lambda: ops.conv3d_roi(x, pc, boxes, x_range=rng))
    assert torch.equal(y_roi, y_full)
    print(f"full launch {t_full:.3f} ms, demand-driven launch listing every tile {t_roi:.3f} ms")
    assert t_roi <= 1.08 * t_full, (t_roi, t_full)


def test_demand_driven_plan_never_reads_what_it_did_not_compute(dev, M):
    """The plan's workspace is pre-filled with NaN: if the final warp (or anything else) read a voxel the demand-driven tail
    skipped, the output would carry it.  Same bits as the full evaluation, reference-like fields and a batch that mixes them
    with frames whose C2D field travels through the whole volume (box = volume: every tile is computed)."""
    from megaportrait_hack_amd import plan as P

    hot = _hot(M, dev)
    b, d, h, w = 3, 16, 32, 32
    inp = {k: v.to(dev) for k, v in R.seeded_hot_inputs(b, 13, D=d, H=h, W=w).items()}
    with torch.no_grad():
        # make frame 1's driver field travel: a large translation pushes its sample positions across the volume
        inp["td"][1] = torch.tensor([9.0, -7.0, 11.0], device=dev)
        inp["Rd"][2] = torch.tensor([170.0, 95.0, -120.0], device=dev)
        full = P.HotSlicePlan(hot, dims=(96, d, h, w), full_final_conv=True)
        lazy = P.HotSlicePlan(hot, dims=(96, d, h, w))
        want = full(**inp)
        for key, ws in list(lazy._ws.items()):
            ws.fill_(0xFF)
        nbytes = lazy.lib.mphip_hot_slice_workspace_bytes(lazy._handle, b)
        poisoned = lazy._workspace("slice", b, nbytes)
        poisoned.view(torch.float32)[: nbytes // 4].fill_(float("nan"))
        got = lazy(**inp)
        ref = hot._run_python(check_shape=False, **inp)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want) and torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(2, 96, 8, 16, 32), (1, 96, 16, 64, 64)])
def test_demand_driven_conv_backward_equals_the_full_backward(dev, shape):
    """When dy is zero outside per-frame boxes (the gradient of a gather), mphip_conv3d_bwd_data_roi / _bwd_weight_roi give what
    the full kernels give on that dy: dx bit for bit (zero-filled outside the grown boxes), dW to rounding (same products, the
    all-zero tiles are skipped)."""
    from megaportrait_hack_amd import ops

    n, c, d, h, w = shape
    gen = torch.Generator().manual_seed(9)
    conv = torch.nn.Conv3d(c, 96, 3, padding=1).to(dev)
    x = torch.randn(n, c, d, h, w, generator=gen).to(dev)
    boxes = _rand_boxes(n, d, h, w, gen)
    dy = torch.zeros(n, 96, d, h, w)
    for i, (lx, ly, lz, ex, ey, ez, _, _) in enumerate(boxes.tolist()):
        dy[i, :, lz:lz + ez, ly:ly + ey, lx:lx + ex] = torch.randn(96, ez, ey, ex, generator=gen) * 1e-3
    dy, boxes = dy.to(dev), boxes.to(dev)
    pc_t = ops.PackedConv(conv.weight.detach(), None, transposed=True)
    _, scale = ops.grad_prep(dy, want_bias=False)
    dx_full = ops.conv3d_bwd_data(dy, pc_t, scale)
    dx_roi = ops.conv3d_bwd_data(dy, pc_t, scale, roi=boxes)
    assert torch.equal(dx_roi, dx_full)
    rng = ops.absmax_range(x)
    dw_full = ops.conv3d_bwd_weight(x, dy, 3, scale, x_range=rng)
    dw_roi = ops.conv3d_bwd_weight(x, dy, 3, scale, x_range=rng, roi=boxes)
    assert torch.equal(dw_roi, dw_full)


@pytest.mark.parametrize("D", [16, 8, 4])
def test_fused_field_coords_equal_compose_plus_coords_bitwise(dev, D):
    """mphip_warp_field_coords (K1 + the coordinate pass in one kernel, no field in HBM) gives exactly the coordinates of
    mphip_warp_field_compose -> mphip_warp_coords — the index contract (bit-exact flow-field index computation) is unchanged."""
    import ctypes

    from megaportrait_hack_amd import _lib, ops

    lib = _lib.load()
    P = ctypes.c_void_p
    B, G = 3, 64
    theta = (R.seeded_tensor((B, 3, 4), 71) * 0.8).to(dev)
    em = R.seeded_tensor((B, 3, 16, 16, 16), 72).abs().to(dev)
    field = ops.warp_field_compose(theta, em, G)
    want = ops.warp_coords(field, D, G, G)
    got = torch.full_like(want, float("nan"))
    _lib.check(lib.mphip_warp_field_coords(P(theta.data_ptr()), P(em.data_ptr()), P(ops.affine_base_table(G, dev).data_ptr()),
                                           P(ops.linspace_table(D, dev).data_ptr()), P(ops.linspace_table(G, dev).data_ptr()),
                                           P(ops.linspace_table(G, dev).data_ptr()), P(got.data_ptr()), B, 16, 16, 16, G, D,
                                           P(torch.cuda.current_stream().cuda_stream)), "mphip_warp_field_coords")
    assert torch.equal(got, want)
