"""Stage-level parity on the CPU model (tests/wavesim): every stage entry point of the C ABI -- which call the very device
functions the production kernels call (codec_launch.inl: debug_stage_kernel for the f32 encode stages and both decode
stages, debug_stage_wide_kernel for the f64 encode stages) -- against the oracle, one hypercube at a time.  Mirrors
src/test/codec_profile_test.inl:514-549 (flattening + forward transform), :552-729 (residual encoding, crafted sparse
pattern), :735-801 (chunk decoding), :889-947 (inverse transform) and codec_generic_test.cc:65-81 (transpose involution).
The same cases run on the GPU in tests/test_hip_stages.py."""
import numpy as np
import pytest

from ndzip_amd import hip
from oracle import oracle
from tests.util import PROFILES, SIDE, profile_id, random_bits, random_unit_floats, sparse_residuals, word_dtype
from tests.wavesim import sim

FWD, ENC, DEC, INV, TR, TRG = 0, 1, 2, 3, 4, 5


@pytest.fixture(autouse=True)
def _model():
    with sim.active():
        yield


def _p(a):
    return a.ctypes.data


@pytest.mark.parametrize("stage", [TR, TRG])
def test_transpose32_matches_oracle_and_is_involution(stage):
    rng = np.random.default_rng(1)
    n = 256
    x = rng.integers(0, 2**32, size=(n, 32), dtype=np.uint32)
    x >>= rng.integers(0, 32, size=(n, 1), dtype=np.uint32)  # like codec_generic_test.cc:70-72
    x[0] = 0
    x[1] = 0xFFFFFFFF
    x[2] = np.uint32(1) << np.arange(32, dtype=np.uint32)
    out = np.zeros_like(x)
    hip.debug_stage(stage, np.float32, 1, None, 0, _p(x), _p(out), None, n)
    for i in range(0, n, 7):
        assert np.array_equal(out[i], oracle.transpose_bits(x[i])), i
    back = np.zeros_like(x)
    hip.debug_stage(stage, np.float32, 1, None, 0, _p(out), _p(back), None, n)
    assert np.array_equal(back, x)


def _grid(dtype, dims, aligned, seed):
    side = SIDE[dims]
    if dims == 1:
        shape = (3 * side + (0 if aligned else 5),)
    elif dims == 2:
        shape = (2 * side + 3, 3 * side + (0 if aligned else 7))
    else:
        shape = (2 * side + 1, 2 * side + 2, 3 * side + (0 if aligned else 3))
    return random_bits(shape, dtype, seed)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("aligned", [True, False], ids=["aligned", "unaligned"])
def test_forward_transform_matches_oracle(profile, aligned):
    dtype, dims = profile
    wdt = word_dtype(dtype)
    data = _grid(dtype, dims, aligned, seed=3)
    nhc = oracle.num_hypercubes(data.shape)
    for hc in sorted({0, 1, nhc // 2, nhc - 1}):
        out = np.zeros(4096, dtype=wdt)
        hip.debug_stage(FWD, dtype, dims, data.shape, hc, _p(data), _p(out))
        want = oracle.forward_transform(oracle.load_cube(data, hc), dims)
        assert np.array_equal(out, want), (hc, np.flatnonzero(out != want)[:8])


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("pattern", ["sparse", "random", "zeros", "ones", "single_bits", "dense_chunks"])
def test_residual_encoding_matches_oracle(profile, pattern):
    dtype, dims = profile
    wdt = word_dtype(dtype)
    bits = np.dtype(wdt).itemsize * 8
    if pattern == "sparse":
        res = sparse_residuals(dtype, seed=5)
    elif pattern == "random":
        res = random_bits((4096,), dtype, 6).view(wdt)
    elif pattern == "zeros":
        res = np.zeros(4096, dtype=wdt)
    elif pattern == "ones":
        res = np.full(4096, np.iinfo(wdt).max, dtype=wdt)
    elif pattern == "dense_chunks":
        # chunks that keep every plane next to empty ones: the 16-byte dense path at aligned and unaligned positions
        res = random_bits((4096,), dtype, 8).view(wdt).copy()
        res.reshape(-1, bits)[::3] = 0
        res.reshape(-1, bits)[1::5, :] &= wdt(0xFF)
    else:
        res = (wdt(1) << (np.arange(4096, dtype=wdt) % wdt(bits))).astype(wdt)
        res[::3] = 0
    want = oracle.encode_cube(res)
    out = np.zeros(4096 + 4096 // bits, dtype=wdt)
    length = np.zeros(1, dtype=np.uint32)
    hip.debug_stage(ENC, dtype, dims, None, 0, _p(res), _p(out), _p(length))
    n = int(length[0])
    assert n == len(want)
    assert np.array_equal(out[:n], want)
    # and decode it back (codec_profile_test.inl:735-801)
    stream = np.zeros(4096 + 4096 // bits, dtype=wdt)
    stream[: len(want)] = want
    back = np.zeros(4096, dtype=wdt)
    hip.debug_stage(DEC, dtype, dims, None, 0, _p(stream), _p(back))
    assert np.array_equal(back, res)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("aligned", [True, False], ids=["aligned", "unaligned"])
def test_inverse_transform_matches_oracle(profile, aligned):
    dtype, dims = profile
    wdt = word_dtype(dtype)
    shape = _grid(dtype, dims, aligned, seed=0).shape
    nhc = oracle.num_hypercubes(shape)
    res = random_bits((4096,), dtype, 9).view(wdt)
    want_cube = oracle.inverse_transform(res, dims)
    for hc in sorted({0, nhc - 1}):
        out = np.zeros(shape, dtype=wdt)
        hip.debug_stage(INV, dtype, dims, shape, hc, _p(res), _p(out))
        assert np.array_equal(oracle.load_cube(out.view(dtype), hc), want_cube), hc
        assert np.count_nonzero(out) <= 4096  # nothing outside the hypercube was touched


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_forward_then_inverse_is_identity(profile):
    """block transform is reversible (codec_profile_test.inl:23-34)"""
    dtype, dims = profile
    wdt = word_dtype(dtype)
    shape = (SIDE[dims],) * dims
    data = random_unit_floats(shape, dtype, 11)
    res = np.zeros(4096, dtype=wdt)
    back = np.zeros(4096, dtype=wdt)
    hip.debug_stage(FWD, dtype, dims, shape, 0, _p(data), _p(res))
    hip.debug_stage(INV, dtype, dims, shape, 0, _p(res), _p(back))
    assert np.array_equal(back, data.reshape(-1).view(wdt))


def test_wave_scan_and_sum():
    """The DPP wave scan (row_shr 1/2/4/8 + row_bcast:15/31) and the wave sum (its last lane by v_readlane) on their own, against
    numpy: uint32 wraparound included."""
    
    rng = np.random.default_rng(4)
    n = 64 * 9
    x = rng.integers(0, 2**32, size=n, dtype=np.uint32)
    x[:64] = 1
    x[64:128] = 0xFFFFFFFF
    x[128:192] = np.arange(64, dtype=np.uint32)
    got = np.zeros(n + n // 64, dtype=np.uint32)
    hip.debug_stage(6, np.float32, 1, None, 0, _p(x), _p(got), None, n)
    want = np.concatenate([np.cumsum(x.reshape(-1, 64).astype(np.uint64), axis=1).astype(np.uint32).reshape(-1),
                           x.reshape(-1, 64).astype(np.uint64).sum(axis=1).astype(np.uint32)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("schedule", ["", "reverse", "random:3"])
@pytest.mark.parametrize("cus,n,grid", [(2, 1 << 12, 0), (5, 1 << 14, 0), (8, 1 << 15, 17), (2, 3000, 1)])
def test_lookback_scan_on_its_own(schedule, cus, n, grid):
    """Stage 7 (the production ticket / publish / look-back / release functions alone) over up to 2^15 synthetic tile lengths, with
    the model's workgroups resumed forwards, backwards and in random order: prefix sums against numpy.  The GPU suite runs the
    same stage at 2^20 tiles (tests/test_hip_stages.py)."""
    from tests.test_hip_stages import _lookback_case

    x = _lookback_case(n, 11)
    out = np.full(n + 2, 0xFFFFFFFF, dtype=np.uint32)
    with sim.active(cus=cus, blocks_per_cu=2, schedule=schedule):
        hip.debug_stage(7, np.float32, 1, None, grid, _p(x), _p(out), None, n)
    incl = np.cumsum(x.astype(np.uint64))
    assert out[n + 1] == 0 and out[n] == incl[-1]
    assert np.array_equal(out[:n], (incl - x).astype(np.uint32))
