"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/mphip.h
declares, argument validation returns error codes (no GPU work), the nn.Module mirror keeps the
reference's state-dict layout, and the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib():
    from megaportrait_hack_amd import _lib

    _lib.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from megaportrait_hack_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "mphip.h")).read()
    declared = set(re.findall(r"\b(mphip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 27
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in mphip.h but not exported by libmphip.so"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with mphip.h"
    # the library reports the ABI version it was built with; the binding refuses a mismatch with the header
    assert lib.mphip_version() == _lib.header_abi_version() == _lib.EXPECTED_ABI_VERSION >= 3


def test_stale_library_is_refused(lib, monkeypatch):
    from megaportrait_hack_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "header_abi_version", lambda: 999)
    with pytest.raises(RuntimeError, match="MPHIP_ABI_VERSION"):
        _lib.load()


def test_argument_validation_needs_no_gpu(lib):
    assert lib.mphip_warp_volume(None, None, None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, None, 0, None) == -1
    assert b"null pointer" in lib.mphip_last_error()
    assert lib.mphip_conv3d_fwd(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 3, 0, None, 0, None) == -1
    one = ctypes.c_void_p(16)
    assert lib.mphip_conv3d_fwd(one, None, one, None, one, 1, 8, 8, 4, 4, 4, 5, 0, None, 0, None) == -1
    assert b"kernel size" in lib.mphip_last_error()
    assert lib.mphip_conv3d_fwd(one, None, one, None, one, 1, 8, 8, 4, 4, 4, 3, 7, None, 0, None) == -1
    assert lib.mphip_groupnorm_stats(one, one, 1, 30, 8, 32, ctypes.c_float(1e-5), None, 0, None) == -1
    assert lib.mphip_avgpool2(one, one, 1, 3, 4, 4, None) == -1
    # pure host helpers
    assert lib.mphip_packed_weight_bytes(96, 96, 3, 0) == 27 * 96 * 96 * 4
    assert lib.mphip_packed_weight_bytes(3, 32, 3, 0) == 27 * 32 * 32 * 4     # Co padded to 32
    assert lib.mphip_packed_weight_bytes(7, 5, 3, 0) == 27 * 6 * 32 * 4       # Ci padded to even
    assert lib.mphip_packed_weight_bytes(4, 4, 2, 0) == 0
    # f16x3: header + f16 hi/lo planes of the 27 taps + (Ci <= 768: layers that can meet an F(2,3) kernel — up to 384 channels the 4-plane
    # kernels, up to 768 the two-frame mode of G3d's 2x8x8 level, r06) the hi/lo planes of its 9 x 4 transformed taps
    assert lib.mphip_packed_weight_bytes(96, 96, 3, 1) == 16 + 27 * 96 * 96 * 2 * 2 + 36 * 96 * 96 * 2 * 2
    assert lib.mphip_packed_weight_bytes(96, 768, 3, 1) == 16 + 27 * 96 * 768 * 2 * 2 + 36 * 96 * 768 * 2 * 2
    assert lib.mphip_packed_weight_bytes(96, 1536, 3, 1) == 16 + 27 * 96 * 1536 * 2 * 2
    assert lib.mphip_packed_weight_bytes(3, 32, 3, 1) == 0                     # f16x3 needs Co%96, Ci%16
    assert lib.mphip_conv3d_supported(8, 96, 96, 16, 64, 64, 3, 1) == 1
    assert lib.mphip_conv3d_supported(8, 96, 96, 16, 64, 64, 1, 1) == 1     # r02: the k=1 split-f16 GEMM kernel
    assert lib.mphip_conv3d_supported(8, 96, 96, 1, 5, 5, 1, 1) == 0        # ... needs whole 64-voxel wave tiles per sample
    assert lib.mphip_conv3d_supported(8, 32, 3, 16, 16, 16, 3, 1) == 0
    assert lib.mphip_conv3d_supported(8, 32, 3, 16, 16, 16, 3, 0) == 1
    # split-K workspace only for small volumes
    assert lib.mphip_conv3d_workspace_bytes(8, 96, 96, 16, 64, 64, 3, 0) == 0
    assert lib.mphip_conv3d_workspace_bytes(8, 96, 96, 16, 64, 64, 3, 1) == (4100 * 4 + 255) // 256 * 256   # f16x3: a library-computed range descriptor
    assert lib.mphip_conv3d_workspace_bytes(1, 768, 768, 2, 8, 8, 3, 0) > 0
    assert lib.mphip_conv3d_workspace_bytes(1, 768, 768, 2, 8, 8, 3, 1) > 0
    assert lib.mphip_groupnorm_workspace_bytes(2, 96, 65536, 32) == 2 * 32 * 12 * 16
    assert lib.mphip_warp_workspace_bytes(8, 16, 64, 64) == 8 * 65536 * 12 + 8 * 16 * 2 * 4   # coordinates + one int per 32x64 K2 tile
    assert lib.mphip_warp_corner_image_bytes(8, 96) == 8 * 6 * 216 * 18 * 4                # K2's optional corner image: [frame][group][6^3][channels + 2], largest grouping (6 x 16 channels)
    one_ = ctypes.c_void_p(16)
    assert lib.mphip_warp_volume_dsum(one_, one_, one_, one_, one_, one_, 1, 2, 4, 4, 4, 4, 4, 4, None, 0, None) == -3   # workspace too small


def test_module_state_dict_layout_matches_reference():
    from megaportrait_hack_amd import model as M

    man = json.load(open(os.path.join(GOLD, "manifest.json")))["state_dict"]
    hot = M.GbaseHotSlice()
    ours = {k: list(v.shape) for k, v in hot.state_dict().items()}
    want = {f"{p}.{k}": s for p, d in man.items() for k, s in d.items()}
    assert ours == want
    assert sum(p.numel() for p in hot.G3d.parameters()) == 48564672
    assert sum(p.numel() for p in hot.warp_generator_s2c.parameters()) == 8807113


def test_checkpoint_loading_quirks():
    from megaportrait_hack_amd import model as M
    from oracle import hotpath_ref as R

    sd = R.seeded_gbase_hot_state_dict(7)
    hot = M.GbaseHotSlice()
    missing, tolerated = M.load_hot_state_dict(hot, sd)
    assert not missing and not tolerated
    assert torch.equal(hot.G3d.final_conv.weight, sd["G3d.final_conv.weight"])
    # a GPU-built reference checkpoint has no adaptive_matrix_* keys (model.py:934-935 quirk)
    gpu_built = {k: v for k, v in sd.items() if "adaptive_matrix" not in k}
    missing, tolerated = M.load_hot_state_dict(M.GbaseHotSlice(), gpu_built)
    assert not missing and len(tolerated) == 4
    # extra keys of a full Gbase checkpoint (Eapp, G2d, ...) are ignored
    full = dict(sd, **{"G2d.conv1.weight": torch.zeros(1), "appearanceEncoder.conv.weight": torch.zeros(1)})
    M.load_hot_state_dict(M.GbaseHotSlice(), full)
    with pytest.raises(KeyError):
        M.load_hot_state_dict(M.GbaseHotSlice(), {k: v for k, v in sd.items() if "final_conv" not in k})


def test_half_products_flag_is_thread_local(lib):
    """mphip_conv3d_set_half_products (the autocast policy switch) returns the previous value and is per thread."""
    import threading

    assert lib.mphip_conv3d_set_half_products(1) == 0
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.mphip_conv3d_set_half_products(0)))
    t.start(); t.join()
    assert seen == [0]                       # another thread starts with the flag off
    assert lib.mphip_conv3d_set_half_products(0) == 1
    assert lib.mphip_conv3d_set_half_products(0) == 0


def test_ablated_build_is_marked_and_refused(lib, tmp_path):
    """csrc/mphip_ablate.h: the product library reports no build flag; a variant with one translation unit compiled under a timing-only
    ablation carries the marker, and the loader refuses it unless MPHIP_ALLOW_ABLATED=1."""
    import subprocess
    import sys

    assert lib.mphip_build_flags() == 0
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "hosttest_abl", "warp", "-DMPHIP_K2_ABL_NOSTORE"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    variant = os.path.join(ROOT, "build_variants", "libmphip_hosttest_abl.so")
    code = "from megaportrait_hack_amd import _lib; print('flags', _lib.load().mphip_build_flags())"
    env = dict(os.environ, MPHIP_LIB=variant, PYTHONPATH=ROOT)
    env.pop("MPHIP_ALLOW_ABLATED", None)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "timing-only ablation" in r.stderr, (r.stdout, r.stderr[-1500:])
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, MPHIP_ALLOW_ABLATED="1"), timeout=300)
        assert r.returncode == 0 and "flags 1" in r.stdout, (r.stdout, r.stderr[-1500:])
    finally:
        os.remove(variant)


def test_pack_cache_rule_inside_and_outside_a_capture():
    """ops.pack_is_current (the cache test of model._PackCache / autograd._bwd_pack): outside a repack_always region a matching key is
    enough; inside one (a hipGraph capture) only a pack that a PackTable run of THAT region wrote counts — a pack made before the
    capture, or by a table run of an earlier capture, must be re-made inside it."""
    from megaportrait_hack_amd import ops

    class Pack:
        _table_token = -1

    pc, key = Pack(), ("ptr", 3)
    assert not ops.pack_is_current(None, key)
    assert not ops.pack_is_current((("ptr", 2), pc), key)          # stale key
    assert ops.pack_is_current((key, pc), key)
    with ops.repack_always():
        assert ops.repacking() and not ops.pack_is_current((key, pc), key)
        pc._table_token = ops._repack_token                        # what PackTable.run() does
        assert ops.pack_is_current((key, pc), key)
    assert not ops.repacking() and ops.pack_is_current((key, pc), key)
    with ops.repack_always():                                      # a later capture: the earlier table run does not count
        assert not ops.pack_is_current((key, pc), key)


def test_no_cpu_fallback():
    from megaportrait_hack_amd import model as M, ops

    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.warp_volume(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        with torch.no_grad():
            M.G3d(96)(torch.zeros(1, 96, 8, 8, 8))
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            M.apply_warping_field(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "megaportrait-hack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "hotpath_c" not in txt, f


def test_captured_tables_match_this_host_or_warn():
    from megaportrait_hack_amd import ops

    cap = ops._captured_tables()
    assert set(cap["linspace"]) == {16, 64} and set(cap["affine_base"]) == {64}
    for n, t in cap["linspace"].items():
        assert t[0] == -1.0 and t[-1] == 1.0 and (t[1:] > t[:-1]).all()
        assert (t - torch.linspace(-1, 1, n)).abs().max() < 2e-7   # same values up to the host's last-bit choices


def test_c_abi_compiles_and_links_from_plain_c(c_abi_exe):
    """gcc (C99, no C++/torch) against include/mphip.h + libmphip.so: the boundary is usable from plain C."""
    assert os.path.isfile(c_abi_exe)


def test_plan_c_program_compiles_and_links(plan_c_exe):
    """tests/c_abi/plan_smoke.c (the one-call entries + index-contract ops from C99) builds against the header and links
    against libmphip.so without a GPU; tests/test_gpu_plan.py runs it."""
    assert os.path.isfile(plan_c_exe)


def test_bench_self_launches_ranks():
    """`python bench.py --gpus N` with no torchrun environment must start N ranks itself (the driver's
    SCALE command).  --dry-launch shows the command; --stub-worker runs the real launch path on CPU over gloo."""
    import json
    import subprocess
    import sys

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "3", "--dry-launch"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and "--dry-launch" not in cmd
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--stub-worker"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"stub_worker": True, "ranks": 2, "n_gpus": 2, "sum": 3.0}]   # exactly ONE line, from rank 0


def test_hot_kernels_have_no_scratch():
    """the kernels the default schedule launches on the hot path must not touch scratch memory — a spilled
    register in a persistent MFMA kernel is a round trip to HBM-backed memory per tile.  Compiles the four hot translation
    units to gfx950 assembly (no GPU needed, ~1 minute) and reads hipcc's kernel metadata (tools/register_table.py; the
    per-round table is profiles/rNN_register_tables.json).  One documented exception: the direct kernel's 512-voxel-tile
    instantiation <4,8,16,8,1> — since r04 a fallback (layers the F(2,3) kernel takes no longer reach it: Ci <= 256 on launches
    that fill the chip) whose 60-odd spilled registers sit in its per-tile prologue / epilogue; its scratch must not grow."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import register_table

    table = register_table.collect(["conv3d_f16x3_wino.hip", "conv3d_f16x3_wino_pp.hip", "conv3d_f16x3_wino_bt.hip", "conv3d_f16x3.hip", "conv3d_bwd_f16x3.hip",
                                    "warp.hip"])
    kernels = {k["demangled"]: k for t in table.values() for k in t["kernels"]}
    hot = [n for n in kernels if any(s in n for s in ("conv3d_k3_f16x3_wino_kernel", "conv3d_k3_f16x3_kernel", "conv3d_k3_f16x3_third_kernel",
                                                       "conv3d_k1_f16x3_kernel", "conv_bwd_weight_f16x3_kernel", "conv_bwd_weight_k1_f16x3_kernel",
                                                       "warp_gather", "warp_coords_kernel", "warp_field_coords_kernel"))]
    assert len(hot) >= 12, sorted(kernels)
    assert not any("conv3d_k3_f16x3_kernel<4, 8, 16, 8, 1>" in n for n in kernels)   # (the r02-r04 fallback with 248-256 B of scratch: removed in r05)
    for n in hot:
        assert kernels[n].get("private_segment_fixed_size", 0) == 0 and kernels[n].get("vgpr_spill_count", 0) == 0, (n, kernels[n])
    for name in ("conv3d_k3_f16x3_wino_kernel", "conv3d_k3_f16x3_wino_pp_kernel"):
        wino = next(k for n, k in kernels.items() if name in n)
        assert wino["group_segment_fixed_size"] <= 160 * 1024 and wino["vgpr_count"] <= 256
    # the big-tile kernel (r06: the two-frame mode of G3d's 2x8x8 level; MPHIP_WINO_PP=2): one wave per SIMD = 512 registers, four
    # instantiations (fused input norm x two-frame mode), none with scratch
    bt = [k for n, k in kernels.items() if "conv3d_k3_f16x3_wino_bt_kernel" in n]
    assert len(bt) == 4
    for k in bt:
        assert k["group_segment_fixed_size"] <= 160 * 1024 and k["vgpr_count"] <= 512 and k["max_flat_workgroup_size"] == 256, k
        assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, k


def test_role_split_conv_hand_issued_memory_ops_are_padded_and_unspilled(monkeypatch):
    """conv3d_f16x3_wino_pp.hip issues its halo loads and LDS-DMA pieces as inline assembly and orders them with its own counted
    s_waitcnt: hipcc neither pads hazards inside an asm string nor knows that a destination register is still in flight.
    What the source promises is checked on the disassembly (no GPU needed): (1) every hand-issued vector-memory group opens with
    `s_nop 4` (its scalar operands may come straight from a v_readlane spill reload: VALU-written SGPR -> VMEM needs 5 wait states);
    (2) no register of the kernel is spilled and no scratch is used (a spill of an in-flight destination would save garbage);
    (3) a halo-load destination is written by exactly one load site per team instantiation and never the source of a compiler copy."""
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    monkeypatch.setenv("MPHIP_KEEP_ASM", "1")
    import register_table

    t = register_table.one(os.path.join(register_table.CSRC, "conv3d_f16x3_wino_pp.hip"))
    ks = [k for k in t["kernels"] if "wino_pp_kernel" in k["name"]]
    assert len(ks) == 2                                          # f16x3 and the single-product (autocast) instantiation
    asm = t["asm"]
    for k in ks:
        assert k.get("vgpr_spill_count", 0) == 0 and k.get("private_segment_fixed_size", 0) == 0, k
        start = asm.index(k["name"] + ":")
        body = asm[start:asm.index(".Lfunc_end", start)]
        blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, flags=re.S)
        vmem = [b for b in blocks if "buffer_load" in b or "global_load_lds" in b]
        assert len(vmem) >= 2 * (3 + 9 + 2), len(vmem)          # per team: three halo-load groups, a DMA group per step, the prologue's
        for b in vmem:
            first = [ln.strip() for ln in b.splitlines() if ln.strip()][0]
            assert first == "s_nop 4", b
        # M0 (the LDS-DMA destination base) is written inside the asm statements only; hipcc rejects an "m0" clobber ("reserved register"), so the
        # guarantee that no compiler-generated instruction depends on M0 across a statement is checked here instead
        outside = re.sub(r";;#ASMSTART\n.*?;;#ASMEND", "", body, flags=re.S)
        assert not re.search(r"\bm0\b", outside), [ln for ln in outside.splitlines() if re.search(r"\bm0\b", ln)][:3]
        # the halo loads' destinations inside the tile loop (hipcc marks a block's loop membership in the label comment; the prologue's loads,
        # outside every loop, are followed by a full drain before anything reads them)
        loop_dest, n_loop, in_loop = set(), 0, False
        for ln in body.splitlines():
            if re.match(r"\.LBB\d+_\d+:", ln):
                in_loop = "Loop" in ln
            m = re.match(r"\s+buffer_load_dword(?:x4)? (v\[(\d+):(\d+)\]|v(\d+)),", ln)
            if m and in_loop:
                n_loop += 1
                loop_dest |= set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
        assert n_loop >= 16 and n_loop % 8 == 0, n_loop            # eight loads a unit, two teams (hipcc may unswitch the tile loop)
        assert 20 <= len(loop_dest) <= 40, sorted(loop_dest)       # one 20-register set per team (the teams may share numbers)
        # a compiler copy FROM a loop destination register would be a copy of data that may still be in flight
        bad = [ln for ln in body.splitlines() if re.match(r"\s+v_(mov_b32|mov_b64|pk_mov_b32|accvgpr_write)", ln) and
               any(re.search(r", v%d$" % r, ln.strip()) for r in loop_dest)]
        assert not bad, bad[:5]


def test_big_tile_conv_hand_issued_memory_ops_are_padded_and_counted(monkeypatch):
    """conv3d_f16x3_wino_bt.hip (one wave per SIMD) issues its LDS-DMA pieces and halo loads as inline assembly, ONE per statement, each hung
    behind its own MFMA, and orders them with counted s_waitcnt.  Checked on the disassembly of all four instantiations: (1) every
    hand-issued vector-memory statement opens with `s_nop 4`; (2) no compiler-generated instruction touches M0; (3) no spill, no scratch;
    (4) the step loop issues exactly what the counted waits assume — per nine-step period 54 LDS-DMA pieces and, per role, the halo loads
    of one unit (8 instructions; 9 in the fused two-frame mode: the row's table values) — and no compiler-visible vector load at all (hipcc
    would wait vmcnt(0) for it in the middle of the MFMA stream); (5) the MFMAs of a step are not separated by more than a handful of
    instructions (the single wave's issue slots are what bounds this kernel: profiles/NOTES_r06.md)."""
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    monkeypatch.setenv("MPHIP_KEEP_ASM", "1")
    import register_table

    t = register_table.one(os.path.join(register_table.CSRC, "conv3d_f16x3_wino_bt.hip"))
    ks = [k for k in t["kernels"] if "wino_bt_kernel" in k["name"]]
    assert len(ks) == 4
    asm = t["asm"]
    for k in ks:
        assert k.get("vgpr_spill_count", 0) == 0 and k.get("private_segment_fixed_size", 0) == 0, k
        m = re.search(r"wino_bt_kernelILb([01])ELb([01])E", k["name"])
        fused, d2 = m.group(1) == "1", m.group(2) == "1"
        start = asm.index(k["name"] + ":")
        body = asm[start:asm.index(".Lfunc_end", start)]
        blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, flags=re.S)
        vmem = [b for b in blocks if "buffer_load" in b or "global_load_lds" in b]
        for b in vmem:
            lines = [ln.strip() for ln in b.splitlines() if ln.strip()]
            assert lines[0] == "s_nop 4", b
            assert sum(1 for ln in lines if ln.startswith(("buffer_load", "global_load_lds"))) <= 2, b   # (one per statement; the 4-byte edge loads go in pairs)
        outside = re.sub(r";;#ASMSTART\n.*?;;#ASMEND", "", body, flags=re.S)
        assert not re.search(r"\bm0\b", outside), [ln for ln in outside.splitlines() if re.search(r"\bm0\b", ln)][:3]
        # the step loop: the blocks hipcc marks as loop members that contain MFMAs
        loop_lines, in_loop = [], False
        for ln in body.splitlines():
            if re.match(r"\.LBB\d+_\d+:", ln):
                in_loop = "Loop" in ln
            elif in_loop and ln.startswith("\t") and not ln.strip().startswith((";", ".")):
                loop_lines.append(ln.strip())
        n_mfma = sum(1 for ln in loop_lines if ln.startswith("v_mfma"))
        assert n_mfma == 9 * 36, n_mfma
        first, last = next(i for i, ln in enumerate(loop_lines) if ln.startswith("v_mfma")), max(i for i, ln in enumerate(loop_lines) if ln.startswith("v_mfma"))
        steps = loop_lines[first:last + 1]
        tail = loop_lines[first:]                     # (+ what hangs behind the last MFMA, and the tile's epilogue: stores only)
        # 54 pieces of the nine slabs; the fused 4-plane mode also re-loads its LDS table by LDS-DMA (one wave, up to three pieces)
        assert sum(1 for ln in tail if ln.startswith("global_load_lds")) == 54 + (3 if fused and not d2 else 0)
        n_halo = sum(1 for ln in tail if ln.startswith("buffer_load_dword"))
        assert n_halo == 2 * (9 if (fused and d2) else 8), (k["name"], n_halo)
        assert not any(ln.startswith(("global_load_dword", "flat_load", "scratch_")) for ln in steps)
        gaps, run = [], 0
        for ln in steps:
            if ln.startswith("v_mfma"):
                gaps.append(run); run = 0
            else:
                run += 1
        assert sorted(gaps)[len(gaps) // 2] <= 4 and sum(gaps) / len(gaps) < 6.0, (sorted(gaps)[len(gaps) // 2], sum(gaps) / len(gaps))
