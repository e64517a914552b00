"""The HIP kernels, executed work-item by work-item on the CPU (tests/wavesim: a wave64 functional model that compiles the
product's kernel sources unchanged), against the oracle and the reference-generated golden fixtures -- bit-exact.

This is NOT the GPU parity run (tests/test_hip_*.py, -m gpu, are) and nothing here is a product path: it lets the
GPU-less authoring container check the kernels' LOGIC -- stencil, bit transposes, plane compaction, chunk scans, the
ticket / decoupled look-back protocol between concurrently running workgroups, header and border handling -- after every
kernel edit.  What it cannot check (timing, bank conflicts, cross-XCD coherence) is what the GPU suite and profiles/ are
for.  Cases mirror src/test/codec_profile_test.inl:37-140, :952-1082 at sizes the model runs in seconds."""
import json
import os

import numpy as np
import pytest

from ndzip_amd import hip
from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import PROFILES, SIDE, profile_id, random_bits, random_unit_floats, same_bits
from tests.wavesim import sim

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLDEN, "hashes.json")) as f:
    META = json.load(f)


@pytest.fixture(scope="module", autouse=True)
def _model():
    sim.load()


def _check(data, **kw):
    want = oracle.compress(data)
    got = sim.compress(data, **kw)
    assert len(got) == len(want), (len(got), len(want))
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"first differing words {bad[:8]}"
    back = sim.decompress(want, data.dtype, data.shape)
    assert same_bits(back, data)
    return got


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_single_hypercube(profile):
    dtype, dims = profile
    _check(random_unit_floats((SIDE[dims],) * dims, dtype, 22))


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("n", [0, 1])
def test_zero_hypercubes(profile, n):
    dtype, dims = profile
    data = np.full((n,) * dims, 42, dtype=dtype)
    assert len(_check(data)) == n


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("kind", ["bits", "unit", "synthetic", "zeros"])
def test_many_hypercubes_aligned_and_with_border(profile, kind):
    dtype, dims = profile
    side = SIDE[dims]
    shapes = {1: [(side * 9,), (side * 3 + 17,)], 2: [(side * 3, side * 4), (side * 2 + 1, side * 3 + 3)],
              3: [(side * 2, side * 3, side * 4), (side * 2 + 5, side + 1, side * 3 + 2)]}[dims]
    for i, shape in enumerate(shapes):
        if kind == "bits":
            data = random_bits(shape, dtype, 30 + i)
        elif kind == "unit":
            data = random_unit_floats(shape, dtype, 40 + i)
        elif kind == "zeros":
            data = np.zeros(shape, dtype)
            data.reshape(-1)[::977] = -0.0
        else:
            data = synth_numpy(shape, dtype, seed=50 + i, noise_mask=0xFF)
        _check(data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_more_workgroups_than_tiles_and_many_resident(profile):
    """The ticket / look-back protocol with 12 concurrently resident workgroups (and with a grid larger than the work)."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 40,), 2: (side * 6, side * 7), 3: (side * 3, side * 4, side * 4)}[dims]
    data = synth_numpy(shape, dtype, seed=5, noise_mask=0xFFF)
    _check(data, cus=4, blocks_per_cu=3)
    small = {1: (side * 2,), 2: (side, side * 3), 3: (side, side, side * 2)}[dims]
    _check(synth_numpy(small, dtype, seed=6, noise_mask=0xFF), cus=4, blocks_per_cu=3)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_one_resident_workgroup(profile):
    """A grid of one persistent workgroup walks every tile itself (every look-back finds its own previous tile)."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 5,), 2: (side * 2, side * 3), 3: (side, side * 2, side * 3)}[dims]
    _check(random_unit_floats(shape, dtype, 3), cus=1, blocks_per_cu=1)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("skew", [1, 3])
def test_element_aligned_pointers(profile, skew):
    """Array and stream pointers that are only element-aligned: the 16-byte accesses of the kernels assume no more."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 3 + 5,), 2: (side * 2 + 3, side * 3), 3: (side * 2, side + 3, side * 2 + 1)}[dims]
    data = synth_numpy(shape, dtype, seed=77 + skew, noise_mask=0xFF)
    holder = np.zeros(data.size + 8, dtype=dtype)
    view = holder[skew: skew + data.size].reshape(shape)
    view[...] = data
    want = oracle.compress(data)
    got = sim.compress(view, misalign_words=skew)
    assert np.array_equal(got, want)


def test_f64_odd_hypercube_count_zeroes_header_pad():
    data = np.zeros(3 * 4096, dtype=np.float64)
    s = _check(data)
    assert len(s) == 194 and s[0] == 0x0000008000000040 and s[1] == 0x00000000000000C0
    s2 = _check(random_unit_floats((200, 70), np.float64, 7))  # NHC = 3
    assert np.frombuffer(s2.tobytes(), dtype=np.uint32)[3] == 0


def test_known_answers_from_reference():
    """Known answers captured from the compiled reference."""
    s = sim.compress(np.zeros(4096, np.float32))
    assert len(s) == 129 and s[0] == 0x80 and not s[1:].any()
    s = sim.compress(np.ones(4096, np.float32))
    assert len(s) == 136 and s[0] == 0x87 and s[1] == 0x7F000000 and not s[2:129].any() and (s[129:] == 0x80000000).all()
    s = sim.compress(np.ones(4096, np.float64))
    assert len(s) == 75 and s[0] == 0x4A and s[1] == 0x7FE0000000000000 and (s[65:] == 0x8000000000000000).all()
    b = np.zeros(4099, np.float32)
    b[4096:] = [1, 2, -1]
    s = sim.compress(b)
    assert len(s) == 132 and list(s[-3:]) == [0x3F800000, 0x40000000, 0xBF800000]


@pytest.mark.parametrize("case", META["small_cases"], ids=lambda c: c["name"])
def test_reference_streams(case):
    """Byte-exact (input, stream) pairs produced by the compiled reference (tests/golden/make_golden.py)."""
    vec = np.load(os.path.join(GOLDEN, "vectors.npz"))
    data, want = vec[case["name"] + "__in"], vec[case["name"] + "__stream"]
    got = sim.compress(data)
    assert len(got) == len(want) and np.array_equal(got, want)
    assert same_bits(sim.decompress(want, data.dtype, data.shape), data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_corrupt_header_entries_are_contained(profile):
    """A header entry that points outside the stream: the affected hypercubes decode as zeros and the error word is set
    (NDZIP_HIP_ERR_DEVICE_FAULT from check()) -- the model would crash on a wild read just like the device."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 4,), 2: (side * 2, side * 2), 3: (side, side * 2, side * 2)}[dims]
    data = random_unit_floats(shape, dtype, 9)
    stream = oracle.compress(data).copy()
    header = stream[: 2 if np.dtype(dtype).itemsize == 8 else 4].view(np.uint32)
    header[1] = 0xFFFFFF00
    with pytest.raises(hip.NdzipHipError, match="corrupt stream header"):
        sim.decompress(stream, dtype, shape, bounded=True)
    with pytest.raises(hip.NdzipHipError, match="corrupt stream header"):
        sim.decompress(stream, dtype, shape, bounded=False)
    # a plausible length but an offset past the words the caller vouched for
    stream2 = oracle.compress(data)
    with pytest.raises(hip.NdzipHipError, match="corrupt stream header"):
        with sim.active():
            dec = hip.make_hip_decompressor(dtype, dims)
            out = np.zeros(shape, dtype)
            dec.decompress(stream2.ctypes.data, out.ctypes.data, shape, stream_length_words=len(stream2) - 40)
            dec.check()


@pytest.mark.parametrize("profile", [(np.float32, 3), (np.float64, 2)], ids=profile_id)
def test_host_pointer_offloader_and_pipelined_offloader(profile):
    dtype, dims = profile
    side = SIDE[dims]
    shape = {2: (side * 2 + 5, side * 3), 3: (side * 2, side * 2 + 3, side * 2)}[dims]
    data = random_unit_floats(shape, dtype, 60)
    want = oracle.compress(data)
    with sim.active():
        off = hip.make_hip_offloader(dtype, dims)
        stream = off.compress(data)
        assert np.array_equal(stream, want)
        back, consumed = off.decompress(stream, shape)
        assert consumed == len(stream) and same_bits(back, data)
        with pytest.raises(hip.NdzipHipError, match="corrupt stream header|longer than the given words|shorter"):
            off.decompress(stream[:-3], shape)
        po = hip.HipPipelinedOffloader(dtype, shape, slots=2)
        outs = [np.zeros(hip.compressed_length_bound(dtype, shape), dtype=want.dtype) for _ in range(2)]
        po.submit_compress(0, data, outs[0])
        po.submit_compress(1, data, outs[1])
        for slot in (0, 1):
            words, _ = po.wait(slot)
            assert words == len(want) and np.array_equal(outs[slot][:words], want)
        back2 = np.zeros(shape, dtype)
        po.submit_decompress(0, want, back2)
        words, _ = po.wait(0)
        assert words == len(want) and same_bits(back2, data)
        po.close()


@pytest.mark.parametrize("dtype,shapes", [(np.float32, [(16 * 5, 16 * 2, 16 * 4), (16 * 2, 16 * 2, 16 * 2), (16 * 5, 16 * 2, 16 * 4)]),
                                          (np.float64, [(64 * 3, 64 * 2), (64 * 1, 64 * 2), (64 * 3, 64 * 2)])])
def test_launch_epoch_lives_in_the_scratch_and_starts_over(dtype, shapes):
    """A compress launch takes nothing per-launch from the host: the descriptor epoch is read from the scratch and advanced by the
    last workgroup to leave (what lets a recorded launch be replayed from a hipGraph).  Here a fresh handle's epoch word is set two
    launches below the end of the 30-bit field and the handle is used seven times, larger and smaller extents in turn: the
    launch that ends on epoch 2^30 - 1 wipes the descriptors and hands epoch 1 to the next one; every stream equals the oracle's."""
    with sim.active(cus=3, blocks_per_cu=2):
        comp = hip.make_hip_compressor(dtype, hip.CompressorRequirements(*shapes))
        # (white box, model only -- its "device memory" is the heap: the handle's scratch pointer sits behind {int, int, uint32, stream},
        # the epoch word where the library's own layout constants put it: ndzip_hip_debug_scratch_epoch_offset, a stage hook)
        import ctypes

        scratch = ctypes.c_uint64.from_address(comp._h.value + 24).value
        offset = hip.stages_lib().ndzip_hip_debug_scratch_epoch_offset()
        assert offset == 16 * 8 + 17 * 128  # (today's layout: 16 reserved descriptors, 16 ticket lines + the 'done' line)
        epoch_word = ctypes.c_uint32.from_address(scratch + offset)
        assert epoch_word.value == 1  # (a fresh handle)
        epoch_word.value = (1 << 30) - 2
        expected = [(1 << 30) - 2, (1 << 30) - 1, 1, 2, 3, 4, 5, 6]
        try:
            for launch in range(7):
                assert epoch_word.value == expected[launch], (launch, epoch_word.value)
                shape = shapes[launch % len(shapes)]
                data = synth_numpy(shape, dtype, seed=40 + launch, noise_mask=0xFFFF if launch % 2 else 0xFF)
                want = oracle.compress(data)
                out = np.zeros(hip.compressed_length_bound(dtype, shape), dtype=want.dtype)
                length = np.zeros(1, dtype=np.uint32)
                comp.compress(data.ctypes.data, shape, out.ctypes.data, length.ctypes.data)
                comp.check()
                assert int(length[0]) == len(want) and np.array_equal(out[: len(want)], want), f"launch {launch}"
            assert epoch_word.value == expected[7]
        finally:
            comp.close()


def test_lookback_timeout_stays_in_bounds_and_is_reported():
    """A look-back that gives up (library variant with a spin limit of 0: any wait for a predecessor is a timeout) after a
    LARGER earlier launch on the same handle, whose descriptors are still in the scratch with another epoch: the partial
    prefix must consist of published lengths only (no write past the stream bound, no wild address), the error word must be
    set (check() fails) and the stream length must be poisoned to 0 for callers that never check."""
    sim.load("spin0", defines=("NDZIP_LOOKBACK_SPIN_LIMIT=0",))
    big = synth_numpy((16 * 6, 16 * 4, 16 * 4), np.float32, seed=1, noise_mask=0xFFFF)
    small = synth_numpy((16 * 2, 16 * 4, 16 * 4), np.float32, seed=2, noise_mask=0xFFFF)
    want = oracle.compress(small)
    saw_timeout = False
    with sim.active(cus=4, blocks_per_cu=3, variant="spin0"):
        comp = hip.make_hip_compressor(np.float32, hip.CompressorRequirements(big.shape))
        bound_big = hip.compressed_length_bound(np.float32, big.shape)
        bound = hip.compressed_length_bound(np.float32, small.shape)
        for attempt in range(20):
            out_big = np.zeros(bound_big, dtype=np.uint32)
            length = np.zeros(1, dtype=np.uint32)
            comp.compress(big.ctypes.data, big.shape, out_big.ctypes.data, length.ctypes.data)
            try:
                comp.check()
            except hip.NdzipHipError:
                pass
            canary = 0xDEADBEEF
            out = np.full(bound + 4096, canary, dtype=np.uint32)
            length[0] = 12345
            comp.compress(small.ctypes.data, small.shape, out.ctypes.data, length.ctypes.data)
            assert (out[bound:] == canary).all(), "a write went past the caller's stream buffer"
            try:
                comp.check()
                assert int(length[0]) == len(want) and np.array_equal(out[: len(want)], want)  # no timeout: a correct stream
            except hip.NdzipHipError as e:
                assert "look-back timeout" in str(e)
                assert int(length[0]) == 0, "a timed-out launch must poison the stream length"
                saw_timeout = True
                break
        comp.close()
    assert saw_timeout, "the spin-limit-0 variant never had to wait for a predecessor in 20 attempts"


@pytest.mark.parametrize("dtype,shape,limit", [(np.float32, (16 * 7 + 3, 32, 16), 16 * 2 * 32 * 16 + 100), (np.float64, (64 * 5, 64 * 2), 64 * 64 * 2 + 1),
                                              (np.float32, (4096 * 6 + 77,), 4096 * 2 + 5), (np.float32, (48, 32, 32), 0)])
def test_chunked_interface_for_arrays_beyond_the_format_limits(dtype, shape, limit):
    """ndzip_hip_chunked_*: dimension 0 cut into slabs of whole hypercube rows, one independent stream per slab, concatenated
    (the reference tool's multi-array file format) -- with the element limit lowered so that small arrays need several slabs."""
    data = synth_numpy(shape, dtype, seed=21, noise_mask=0xFFF)
    with sim.active():
        rows, n, bound = hip.chunked_plan(dtype, shape, limit)
        side = SIDE[len(shape)]
        rest = int(np.prod(shape[1:], dtype=np.int64))
        if limit == 0:
            assert (rows, n) == (shape[0], 1)
        else:
            assert rows % side == 0 and rows * rest <= limit and n == -(-shape[0] // rows) and n > 1
        want = np.concatenate([oracle.compress(data[k * rows: (k + 1) * rows]) for k in range(n)])
        assert bound >= len(want)
        got = hip.chunked_compress(data, limit)
        assert len(got) == len(want) and np.array_equal(got, want)
        back, consumed = hip.chunked_decompress(want, dtype, shape, limit)
        assert consumed == len(want) and same_bits(back, data)
        with pytest.raises(hip.NdzipHipError):
            hip.chunked_decompress(want[:-1], dtype, shape, limit)


def test_chunked_plan_at_the_real_limits():
    """Host-only arithmetic: 16 GiB of float32 (2^32 elements) and larger extents split into legal slabs."""
    with sim.active():
        rows, n, bound = hip.chunked_plan(np.float32, (4096, 1024, 1024), 0)  # 2^32 elements: one too many for index_type
        assert n == 2 and rows % 16 == 0 and rows * 1024 * 1024 < 2 ** 32 and rows * n >= 4096
        assert bound < 2 * 2 ** 32
        rows, n, _ = hip.chunked_plan(np.float64, (1 << 36,), 0)
        assert rows % 4096 == 0 and rows < 2 ** 32 and n == -(-(1 << 36) // rows)
        rows, n, _ = hip.chunked_plan(np.float32, (2048, 1024, 1024), 0)  # 2^31 elements fit
        assert (rows, n) == (2048, 1)
        with pytest.raises(hip.NdzipHipError):
            hip.chunked_plan(np.float32, (64, 1 << 20, 1 << 20), 0)  # one row of hypercubes is already too large


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("schedule", ["reverse", "random:1", "random:2", "random:3"])
def test_result_does_not_depend_on_the_order_work_items_run_in(profile, schedule):
    """Between two synchronisation points the model runs a workgroup's work-items one after the other; a kernel without
    intra-workgroup races gives the same stream whatever that order is (forward in every other test; reversed and shuffled
    here).  An order-dependent result = a missing barrier."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 7 + 3,), 2: (side * 3, side * 2 + 5), 3: (side * 2, side * 2 + 1, side * 2)}[dims]
    data = synth_numpy(shape, dtype, seed=31, noise_mask=0xFFF)
    want = oracle.compress(data)
    got = sim.compress(data, cus=3, blocks_per_cu=2, schedule=schedule)
    assert len(got) == len(want) and np.array_equal(got, want)
    assert same_bits(sim.decompress(want, dtype, shape, schedule=schedule), data)
    if np.dtype(dtype).itemsize == 8:  # (the default above is the 256-work-item decoder; the 128-work-item one stays selectable)
        assert same_bits(sim.decompress(want, dtype, shape, schedule=schedule, f64_work_items=128), data)


@pytest.mark.parametrize("shape", [(32, 32, 48), (16, 48, 16), (48, 16, 80)])
def test_f32_3d_with_an_odd_hypercube_count_along_x_takes_the_unpaired_loads(shape):
    """Aligned rows but an odd number of hypercubes along x: tiles 2m, 2m+1 are not x-neighbours everywhere, so the launcher
    must pick the unpaired kernel (compress_kernel_db<float, 3, true, false>)."""
    _check(synth_numpy(shape, np.float32, seed=41, noise_mask=0xFFF), cus=3, blocks_per_cu=2)


@pytest.mark.parametrize("resident", [16, 20])
def test_sixteen_ticket_classes_with_a_partly_resident_grid(resident, monkeypatch):
    """The same with 16 ticket classes (grids of 16+ workgroups): what the scheme needs is one live drawer per class, i.e. the
    first 16 workgroups resident (docs/rounds.md section 4, "Tile order = ticket order"); workgroups 16.. of the 24 may start late or never."""
    monkeypatch.setenv("WAVESIM_MAX_RESIDENT", str(resident))
    for dtype, shape in ((np.float32, (6 * 16, 4 * 16, 4 * 16)), (np.float64, (7 * 64, 6 * 64 + 5))):
        data = synth_numpy(shape, dtype, seed=10)
        want = oracle.compress(data)
        got = sim.compress(data, cus=8, blocks_per_cu=3)
        assert len(got) == len(want) and np.array_equal(got, want)


@pytest.mark.parametrize("resident", [1, 3, 7])
def test_grid_larger_than_what_is_resident(resident, monkeypatch):
    """Tiles are drawn with tickets, never assigned to a block index: a launch whose grid is only partly resident (another
    stream's kernels occupy CUs; here: 24 workgroups, `resident` of them at a time, the others start when those have exited)
    completes with the right stream -- the late workgroups find no ticket left and leave.  A statically assigned first tile
    would hang here (its successors' look-back waits for a workgroup that cannot start)."""
    monkeypatch.setenv("WAVESIM_MAX_RESIDENT", str(resident))
    for dtype, shape in ((np.float32, (3 * 16, 2 * 16, 4 * 16)), (np.float64, (5 * 64 + 3, 3 * 64))):
        data = synth_numpy(shape, dtype, seed=9)
        want = oracle.compress(data)
        got = sim.compress(data, cus=8, blocks_per_cu=3)
        assert len(got) == len(want) and np.array_equal(got, want)
        assert same_bits(sim.decompress(want, dtype, shape), data)


def test_host_pointer_paths_relaunch_once_after_a_lookback_timeout(tmp_path):
    """ndzip_hip_offload_compress / ndzip_hip_offloader_wait: a launch whose look-back timed out is repeated once (the input
    is still on the device) before the caller sees NDZIP_HIP_ERR_DEVICE_FAULT.  Driven on the spin-limit-0 variant (any wait for
    a predecessor is a timeout) in a subprocess with NDZIP_VERBOSE set, so that the relaunches can be counted on stderr: every
    call either returns the oracle's stream or fails with "again after one relaunch", and never more than one relaunch per call."""
    import subprocess
    import sys

    script = tmp_path / "drive.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from ndzip_amd import hip\n"
        "from ndzip_amd.synth import synth_numpy\n"
        "from oracle import oracle\n"
        "from tests.wavesim import sim\n"
        "sim.load('spin0', defines=('NDZIP_LOOKBACK_SPIN_LIMIT=0',))\n"
        "data = synth_numpy((16 * 6, 16 * 4, 16 * 4), np.float32, seed=1, noise_mask=0xFFFF)\n"
        "want = oracle.compress(data)\n"
        "ok = failed = 0\n"
        "with sim.active(cus=4, blocks_per_cu=3, variant='spin0'):\n"
        "    off = hip.make_hip_offloader(np.float32, 3)\n"
        "    po = hip.HipPipelinedOffloader(np.float32, data.shape, slots=1)\n"
        "    out = np.zeros(hip.compressed_length_bound(np.float32, data.shape), dtype=np.uint32)\n"
        "    for i in range(6):\n"
        "        print('CALL', file=sys.stderr, flush=True)\n"
        "        try:\n"
        "            if i % 2 == 0:\n"
        "                got = off.compress(data)\n"
        "            else:\n"
        "                po.submit_compress(0, data, out)\n"
        "                words, _ = po.wait(0)\n"
        "                got = out[:words]\n"
        "            assert np.array_equal(got, want)\n"
        "            ok += 1\n"
        "        except hip.NdzipHipError as e:\n"
        "            assert 'again after one relaunch' in str(e), str(e)\n"
        "            failed += 1\n"
        "    po.close()\n"
        "print('RESULT', ok, failed)\n")
    env = dict(os.environ, NDZIP_VERBOSE="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ok, failed = (int(x) for x in r.stdout.split("RESULT")[1].split())
    assert ok + failed == 6
    calls = r.stderr.split("CALL\n")[1:]
    assert len(calls) == 6
    relaunches = [c.count("relaunching once") for c in calls]
    assert max(relaunches) <= 1, "more than one relaunch in a call"
    assert sum(relaunches) >= failed, "a call failed without having been relaunched"
    assert sum(relaunches) > 0, "the spin-limit-0 variant never timed out in 6 calls"
