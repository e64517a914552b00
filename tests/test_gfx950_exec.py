"""The BUILT gfx950 code objects, executed on the CPU (tests/gfx950_exec.py: an instruction-level interpreter, 64 lanes per
wavefront, several workgroups in flight) against the oracle -- bit-exact.

The functional model (tests/wavesim) compiles the kernels' C++ for the host: it checks their logic, not what hipcc made of it, and it
replaces the one file with hand-written assembly.  Here the model only plays the HOST side (capi.hip: geometry, launch sequences,
scratch); every kernel launch is intercepted and the kernel of the same mangled name is fetched from the gfx950 ELF inside
ndzip_amd/libndzip_hip.so / libndzip_hip_stages.so and interpreted: the EXEC-masked plane compaction as assembled, every DPP control
word and v_readfirstlane pin as emitted, the compiler's register allocation around the inline assembly, the ticket / look-back
protocol of concurrently running workgroups under different interleavings.  What this cannot see: timing and the memory model
across XCDs -- the GPU suite's job; wait states are only book-kept against the interpreter's own hazard table (the authority for
the inline assembly's wait states is LLVM's hazard recogniser: tests/test_asm_hazards.py).  Not a product path."""
import os

import numpy as np
import pytest

from ndzip_amd import hip
from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests import gfx950_exec as gx
from tests.util import PROFILES, SIDE, profile_id, random_bits, same_bits, sparse_residuals, word_dtype
from tests.wavesim import build as simbuild
from tests.wavesim import sim

pytestmark = pytest.mark.skipif(not os.path.exists(gx.OBJDUMP), reason="llvm-objdump of the ROCm toolchain")


@pytest.fixture(scope="module")
def bridge(tmp_path_factory):
    from ndzip_amd import build

    build.build()
    return gx.Bridge(simbuild.build(), [hip.LIB_PATH, hip.STAGES_LIB_PATH], str(tmp_path_factory.mktemp("gfx950")))


def _mixed(shape, dtype, seed):
    """smooth data with noise, an incompressible stretch, an all-zero stretch: chunks of every density in one array"""
    data = synth_numpy(shape, dtype, seed=seed, noise_mask=0xFFFF).reshape(-1).copy()
    n = data.size
    data[n // 5: n // 5 + n // 7] = random_bits((n // 7,), dtype, seed)
    data[n // 2: n // 2 + n // 9] = 0
    return data.reshape(shape)


def _roundtrip(bridge, data, cus=2, blocks_per_cu=2, **dec):
    want = oracle.compress(data)
    with bridge:
        got = sim.compress(data, cus=cus, blocks_per_cu=blocks_per_cu)
        back = sim.decompress(want, data.dtype, data.shape, **dec)
    assert len(got) == len(want), (len(got), len(want))
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"first differing words {bad[:8]}"
    assert same_bits(back, data)
    names = [n for n, *_ in bridge.launched]
    assert any("compress_kernel" in n and "decompress" not in n for n in names) and any("decompress_kernel" in n for n in names), names
    return names


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_code_object_reproduces_the_oracle(bridge, profile):
    """whole hypercubes, several tiles, more tiles than workgroups in flight"""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 5,), 2: (side * 2, side * 3), 3: (side, side * 2, side * 2)}[dims]
    _roundtrip(bridge, _mixed(shape, dtype, 11 + dims))


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_code_object_with_border_and_unaligned_rows(bridge, profile):
    """extents that are no multiple of the side length: the border kernels, element-aligned rows (the unaligned kernel variants),
    an odd hypercube count (a half-empty last f32 tile; the f64 header pad)"""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 3 + 77,), 2: (side + 3, side * 3 + 5), 3: (side + 1, side * 3 + 2, side + 3)}[dims]
    names = _roundtrip(bridge, _mixed(shape, dtype, 21 + dims))
    assert any("border_kernel" in n for n in names)


def test_f64_decoder_with_128_work_items(bridge):
    for shape in ((16, 16, 32), (64, 128), (4096 * 2,)):
        names = _roundtrip(bridge, _mixed(shape, np.float64, 31), f64_work_items=128)
        assert any("decompress_kernelId" in n for n in names) and not any("decompress_kernel_wide" in n for n in names)


def test_f32_3d_unpaired_tiles(bridge):
    """an odd hypercube count along x: compress_kernel_db<float, 3, true, false>"""
    names = _roundtrip(bridge, _mixed((16, 32, 48), np.float32, 41))
    assert any(n.endswith("Lb1ELb0EEEvPKNS_7word_ofIT_E4typeENS_9grid_geomEPjPS5_PyS9_jS9_jS9_") for n in names), names


@pytest.mark.parametrize("quantum,cus,resident,order", [(37, 8, 32, "forward"), (700, 3, 3, "forward"), (4000, 8, 32, "forward"), (150, 2, 1, "forward"),
                                                        (37, 8, 32, "reverse"), (901, 8, 32, "random:3"), (150, 2, 1, "reverse")])
def test_tickets_and_lookback_under_different_interleavings(bridge, quantum, cus, resident, order, monkeypatch):
    """24 tiles over 16 (24) workgroups, all in flight (sixteen ticket classes: the launcher's grid is what is resident, and the classes rely
    on it), and over 6 / 4 workgroups of which only 3 / 1 are in flight at a time (one class = a single global order, which needs no
    co-residency at all).  The wavefronts in flight are interleaved `quantum` instructions at a time, so aggregates are published
    and windows are read in very different orders from case to case.  `order`: which workgroup / wavefront takes its turn first --
    index order, reverse, reshuffled every pass (a race shows only when its loser runs first; the whole module also runs under
    GFX950_EXEC_ORDER=reverse / random:<seed>, profiles/r06_code_object_rehearsal.txt)."""
    real = gx.run_grid
    monkeypatch.setattr(gx, "run_grid", lambda *a, **k: real(*a, **{**k, "quantum": quantum, "resident": resident, "order": order}))
    data = _mixed((32, 64, 96), np.float32, 51)  # 48 hypercubes = 24 tiles
    want = oracle.compress(data)
    with bridge:
        got = sim.compress(data, cus=cus, blocks_per_cu=2)  # (the occupancy is cached per process: 2 workgroups per "CU" since the first launch)
    assert len(got) == len(want) and np.array_equal(got, want)
    (name, grid, block, _), = [l for l in bridge.launched if "compress_kernel" in l[0]]
    # (workgroups per "CU" = what the process's first launch cached: 2 in this module's order, 3 if another module's test ran first)
    assert grid in (2 * cus, 3 * cus) and block == 256 and (grid >= 16) == (cus == 8)


def test_device_wide_scan_stage(bridge):
    """stage 7 (debug_lookback_kernel in libndzip_hip_stages.so): the production ticket / publish / look-back / release code over
    1 500 synthetic tile lengths, 17 one-wavefront workgroups (sixteen ticket classes), two launches on one scratch"""
    rng = np.random.default_rng(7)
    n = 1500
    x = rng.integers(128, 8449, size=n, dtype=np.uint32)
    out = np.full(n + 2, 0xFFFFFFFF, dtype=np.uint32)
    with bridge:
        with sim.active():
            hip.debug_stage(7, np.float32, 1, None, 17, x.ctypes.data, out.ctypes.data, None, n)
    incl = np.cumsum(x.astype(np.uint64))
    assert out[n + 1] == 0 and out[n] == incl[-1]
    assert np.array_equal(out[:n], (incl - x).astype(np.uint32))
    assert sum("debug_lookback_kernel" in l[0] for l in bridge.launched) == 2


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_stage_kernels(bridge, profile):
    """the encode stage (f32: the EXEC-masked compaction as assembled into the stage kernel) and the decode stages on the crafted
    sparse residual pattern of the reference's stage tests (src/test/codec_profile_test.inl:561-567)"""
    dtype, dims = profile
    wdt = word_dtype(dtype)
    bits = np.dtype(wdt).itemsize * 8
    res = sparse_residuals(dtype, seed=5)
    want = oracle.encode_cube(res)
    out = np.zeros(4096 + 4096 // bits, dtype=wdt)
    length = np.zeros(1, dtype=np.uint32)
    with bridge:
        with sim.active():
            hip.debug_stage(1, dtype, dims, None, 0, res.ctypes.data, out.ctypes.data, length.ctypes.data)
            assert int(length[0]) == len(want) and np.array_equal(out[:len(want)], want)
            stream = np.zeros(4096 + 4096 // bits, dtype=wdt)
            stream[:len(want)] = want
            for stage in ((2, 8) if bits == 64 else (2,)):
                back = np.zeros(4096, dtype=wdt)
                hip.debug_stage(stage, dtype, dims, None, 0, stream.ctypes.data, back.ctypes.data)
                assert np.array_equal(back, res), stage
    assert len(bridge.launched) >= 2


def test_corrupt_header_is_contained(bridge):
    """a header entry the format rules out: the hypercube decodes as zeros, the error word is set, nothing outside is read"""
    data = _mixed((16, 16, 48), np.float32, 61)
    stream = oracle.compress(data).copy()
    stream[1] = 0xFFFFFFF0
    out = np.zeros(data.shape, dtype=data.dtype)
    with bridge:
        with sim.active():
            dec = hip.make_hip_decompressor(np.float32, 3)
            dec.decompress(stream.ctypes.data, out.ctypes.data, data.shape, stream.size)
            with pytest.raises(hip.NdzipHipError):
                dec.check()
            dec.close()
    assert same_bits(out[:, :, :16], data[:, :, :16])  # (hypercube 0 is intact; 1 and 2 depend on the broken entry)


def test_instruction_coverage_is_complete(bridge):
    """every instruction of every codec kernel in the two libraries has semantics here (a new compiler or a new kernel that brings
    an instruction this interpreter does not know fails HERE, by name, not as a wrong stream somewhere)"""
    missing = {}
    for co in bridge.cos:
        for name in co.kernels:
            k = gx.Kernel(co, name)
            if k.missing:
                missing[name[:60]] = k.missing
    assert not missing, missing


def _stage_cases():
    from tests import test_wavesim_stages as st

    cases = [("transpose-perm", st.test_transpose32_matches_oracle_and_is_involution, (st.TR,)),
             ("transpose-generic", st.test_transpose32_matches_oracle_and_is_involution, (st.TRG,)),
             ("wave-scan", st.test_wave_scan_and_sum, ())]
    for p in PROFILES:
        for aligned in (True, False):
            tag = f"{profile_id(p)}-{'aligned' if aligned else 'unaligned'}"
            cases.append((f"forward-{tag}", st.test_forward_transform_matches_oracle, (p, aligned)))
            cases.append((f"inverse-{tag}", st.test_inverse_transform_matches_oracle, (p, aligned)))
        for pattern in ("random", "zeros", "ones", "single_bits", "dense_chunks"):
            cases.append((f"encode-decode-{profile_id(p)}-{pattern}", st.test_residual_encoding_matches_oracle, (p, pattern)))
    return cases


@pytest.mark.parametrize("case", _stage_cases(), ids=lambda c: c[0])
def test_stage_tests_of_the_model_suite_on_the_code_object(bridge, case):
    """tests/test_wavesim_stages.py, function for function, with the stage kernels of libndzip_hip_stages.so interpreted instead of
    compiled for the host: the forward / inverse transforms at the first, middle and last hypercube of aligned and unaligned grids,
    the residual coding patterns (all-ones, single bits, dense chunks next to empty ones ...), both 32x32 bit-transpose networks,
    the DPP wave scan."""
    _, fn, args = case
    with bridge:
        with sim.active():
            fn(*args)
    assert bridge.launched and all("debug_" in n for n, *_ in bridge.launched), bridge.launched


def _codec_cases():
    from tests import test_wavesim_codec as ct

    cases = [(f"golden-{c['name']}", ct.test_reference_streams, (c,)) for c in ct.META["small_cases"]]
    cases += [("known-answers", ct.test_known_answers_from_reference, ()), ("f64-odd-hypercube-count", ct.test_f64_odd_hypercube_count_zeroes_header_pad, ())]
    for p in PROFILES:
        cases.append((f"element-aligned-{profile_id(p)}", ct.test_element_aligned_pointers, (p, 1)))
        cases.append((f"corrupt-header-{profile_id(p)}", ct.test_corrupt_header_entries_are_contained, (p,)))
        cases.append((f"zero-hypercubes-{profile_id(p)}", ct.test_zero_hypercubes, (p, 1)))
    return cases


@pytest.mark.parametrize("dtype,shapes", [(np.float32, [(16 * 3, 16 * 2, 16 * 2), (16 * 1, 16 * 2, 16 * 2)]), (np.float64, [(64 * 2, 64 * 2), (64 * 1, 64 * 1)])])
def test_launch_epoch_of_the_code_object(bridge, dtype, shapes):
    """tests/test_wavesim_codec.py::test_launch_epoch_lives_in_the_scratch_and_starts_over on the interpreted code object: the epoch
    load on the way in, the last workgroup's store of the next one, the wipe of the descriptors when the 30-bit field starts over."""
    from tests import test_wavesim_codec as ct

    with bridge:
        ct.test_launch_epoch_lives_in_the_scratch_and_starts_over(dtype, shapes)
    assert sum("compress_kernel" in n for n, *_ in bridge.launched) == 7, bridge.launched


@pytest.mark.parametrize("case", _codec_cases(), ids=lambda c: c[0])
def test_codec_tests_of_the_model_suite_on_the_code_object(bridge, case):
    """tests/test_wavesim_codec.py, function for function, on the interpreted code object: the byte-exact (input, stream) pairs
    generated by the compiled reference (tests/golden), the reference's known answers, arrays and streams at element-aligned
    addresses, rejected header entries, the f64 header pad, inputs smaller than a hypercube (border kernels only)."""
    _, fn, args = case
    with bridge:
        fn(*args)
    assert bridge.launched


@pytest.mark.parametrize("extent,dtype,world", [((48, 16, 32), np.float32, 3), ((130, 70), np.float64, 2)])
def test_sharded_path_kernels_on_the_code_object(bridge, extent, dtype, world):
    """tests/test_hip_sharded.py::test_every_rank_of_a_plan_on_one_gpu with host tensors: every rank's compress_split, the fused
    base + global-offset kernel (offset_header_gathered_kernel), the slab decodes from device-resident bases -- all interpreted."""
    from tests import test_hip_sharded as sh

    with bridge:
        with sim.active():
            sh.test_every_rank_of_a_plan_on_one_gpu(None, "cpu", extent, dtype, world)
            sh.test_emulated_shards_concatenate_to_the_single_stream(None, "cpu", extent, dtype)
    names = {n for n, *_ in bridge.launched}
    assert any("offset_header_gathered_kernel" in n for n in names) and any("offset_header" in n and "gathered" not in n for n in names), names


def test_hazard_checker_fires_on_what_it_is_meant_to_catch():
    """the wait-state rules of tests/gfx950_exec.py::check_hazards on hand-made sequences: too close = reported, padded = clean
    (the shipped EXEC-masked compaction ends in `s_nop 1` for exactly the v_cmpx -> DPP rule)"""
    def run(lines):
        w = gx.Wave(gx.Workgroup(0, 1, 64), 0, {}, 0)
        before = len(gx.HAZARD_LOG)
        for k, text in enumerate(lines):
            op, _, rest = text.partition(" ")
            args, mods = gx._split_operands(rest)
            gx.check_hazards(w, gx.Ins(op, args, mods, 4 * k, 4, text))
        found = gx.HAZARD_LOG[before:]
        del gx.HAZARD_LOG[before:]
        return found

    dpp = "v_mov_b32_dpp v4, v1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    assert len(run(["v_add_u32_e32 v1, v2, v3", dpp])) == 1
    assert len(run(["v_add_u32_e32 v1, v2, v3", "s_nop 0", dpp])) == 1
    assert run(["v_add_u32_e32 v1, v2, v3", "s_nop 1", dpp]) == []
    tail = ["ds_write_b32 v9, v8", "v_add_u32_e32 v9, 4, v9", "s_mov_b64 exec, s[4:5]"]
    assert len(run(["v_cmpx_ne_u32_e32 vcc, 0, v8"] + tail + [dpp])) == 1          # 3 wait states: two short
    assert run(["v_cmpx_ne_u32_e32 vcc, 0, v8"] + tail + ["s_nop 1", dpp]) == []    # what lds_append_nonzero ends with
    assert len(run(["v_readfirstlane_b32 s6, v0", "s_nop 3", "global_load_dword v1, v2, s[6:7]"])) == 1  # (s_nop N = N + 1 states)
    assert run(["v_readfirstlane_b32 s6, v0", "s_nop 4", "global_load_dword v1, v2, s[6:7]"]) == []
    assert len(run(["v_readfirstlane_b32 s9, v0", "s_nop 2", "v_readlane_b32 s3, v5, s9"])) == 1
    assert run(["v_readfirstlane_b32 s9, v0", "s_nop 3", "v_readlane_b32 s3, v5, s9"]) == []


def test_lookback_timeout_path_of_the_code_object(tmp_path, monkeypatch):
    """tests/test_wavesim_codec.py::test_lookback_timeout_stays_in_bounds_and_is_reported on the interpreted spin-limit-0 build
    (ndzip_amd/_variants/spin0.so): the give-up path as compiled -- partial prefix from published lengths only (the canary behind
    the caller's buffer stays intact), error word through the returning atomic, stream length poisoned by the last workgroup out."""
    from ndzip_amd import build
    from tests import test_wavesim_codec as ct

    build.build_test_variants()
    spin0 = os.path.join(os.path.dirname(hip.LIB_PATH), "_variants", "spin0.so")
    model = simbuild.build(variant="spin0", defines=("NDZIP_LOOKBACK_SPIN_LIMIT=0",))
    real = gx.run_grid
    # an adversarial schedule: workgroup 1 gets 40 instructions per turn, the others 3 000 -- it draws one of the first tickets like
    # everybody else and then crawls, so its tile is published long after the successors looked for it: their look-backs have to
    # wait, which this build turns into a time-out at once
    monkeypatch.setattr(gx, "run_grid", lambda *a, **k: real(*a, **{**k, "quantum": lambda wg: 40 if wg == 1 else 3000}))
    b = gx.Bridge(model, [spin0], str(tmp_path))
    with b:
        ct.test_lookback_timeout_stays_in_bounds_and_is_reported()
    assert sum("compress_kernel_db" in n for n, *_ in b.launched) >= 2


def test_waitcnt_checker_fires_on_what_it_is_meant_to_catch():
    """the s_waitcnt bookkeeping of tests/gfx950_exec.py::check_waits on hand-made sequences: a loaded register used before the counter
    says it has arrived, a scalar load consumed under lgkmcnt(1), a barrier crossed with LDS stores in flight -- reported; the
    properly waited forms -- clean (in-order return: vmcnt(1) frees the older of two loads)"""
    def run(lines):
        w = gx.Wave(gx.Workgroup(0, 1, 64), 0, {}, 0)
        before = len(gx.WAIT_LOG)
        for k, text in enumerate(lines):
            op, _, rest = text.partition(" ")
            args, mods = gx._split_operands(rest)
            gx.check_waits(w, gx.Ins(op, args, mods, 4 * k, 4, text))
        found = gx.WAIT_LOG[before:]
        del gx.WAIT_LOG[before:]
        return found

    load1, load2 = "global_load_dword v1, v2, s[0:1]", "global_load_dwordx4 v[4:7], v2, s[0:1] offset:16"
    assert len(run([load1, "v_add_u32_e32 v3, v1, v1"])) == 1
    assert run([load1, "s_waitcnt vmcnt(0)", "v_add_u32_e32 v3, v1, v1"]) == []
    assert run([load1, load2, "s_waitcnt vmcnt(1)", "v_add_u32_e32 v3, v1, v1"]) == []           # the older load has returned
    assert len(run([load1, load2, "s_waitcnt vmcnt(1)", "v_add_u32_e32 v3, v6, v1"])) == 1       # ... the younger one has not
    assert len(run([load1, "v_mov_b32_e32 v1, 0"])) == 1                                          # overwriting a register a load will write
    assert len(run(["ds_read_b32 v8, v9", "s_load_dword s4, s[0:1], 0x0", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v3, v8"])) == 1  # (scalar loads return in any order)
    assert run(["ds_read_b32 v8, v9", "ds_read_b32 v10, v9 offset:4", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v3, v8"]) == []
    assert len(run(["s_load_dwordx2 s[4:5], s[0:1], 0x8", "s_add_u32 s6, s4, 1"])) == 1
    assert len(run(["ds_write_b32 v1, v2", "s_barrier"])) == 1                                    # what lds_append_complete() is there for
    assert run(["ds_write_b32 v1, v2", "s_waitcnt lgkmcnt(0)", "s_barrier"]) == []


def test_no_wait_state_hazard_in_any_executed_stream():
    """(last in this module) every instruction the tests above executed went through the hazard rules: nothing reported -- in
    particular not at the boundaries of the inline assembly, which the compiler's own hazard recogniser cannot see into"""
    assert gx.HAZARD_LOG == [], gx.HAZARD_LOG[:5]
    # ... and through the s_waitcnt bookkeeping: no register used before its load was waited for, no barrier crossed with LDS stores
    # in flight (the compaction's ds_write_b32 sit inside an asm statement: the compiler does not count them)
    assert gx.WAIT_LOG == [], gx.WAIT_LOG[:5]
