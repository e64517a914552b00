"""Stage-level parity: every stage of the HIP path against the CPU oracle, one hypercube at a time, through
the C ABI's stage entry points.  Mirrors the reference's stage tests (src/test/codec_profile_test.inl):
  hypercube flattening + forward transform identical CPU vs GPU   :514-549, :889-947
  residual encodings identical, with the crafted sparse pattern   :552-729
  GPU chunk decoding of a CPU-encoded cube                        :735-801
  inverse transform identical CPU vs GPU                          :889-947
and the bit-transpose involution test of src/test/codec_generic_test.cc:65-81.
The stage entry points call the device functions the production kernels call (codec_launch.inl: debug_stage_kernel = the
f32 encode stages of compress_kernel_db and the decode stages of decompress_kernel; debug_stage_wide_kernel = the f64
encode stages of compress_kernel_wide).  Integer/bit work: the bar is bit-exact."""
import numpy as np
import pytest

from oracle import oracle
from tests.util import PROFILES, SIDE, profile_id, random_bits, random_unit_floats, sparse_residuals, word_dtype

pytestmark = pytest.mark.gpu

FWD, ENC, DEC, INV, TR, TRG = 0, 1, 2, 3, 4, 5
DEC_WIDE, INV_WIDE = 8, 9  # stages 2 / 3 through the 256-work-item decoder of the 64-bit profiles


def _t(a, device):
    import torch

    a = np.ascontiguousarray(a)
    it = np.int32 if a.dtype.itemsize == 4 else np.int64
    return torch.from_numpy(a.reshape(-1).view(it)).to(device)


def _np(t, wdt):
    return t.cpu().numpy().view(wdt)


@pytest.mark.parametrize("stage", [TR, TRG])
def test_transpose32_matches_oracle_and_is_involution(hiplib, cuda_device, stage):
    import torch

    from ndzip_amd import hip

    rng = np.random.default_rng(1)
    n = 4096
    x = rng.integers(0, 2**32, size=(n, 32), dtype=np.uint32)
    x >>= rng.integers(0, 32, size=(n, 1), dtype=np.uint32)  # like codec_generic_test.cc:70-72
    x[0] = 0
    x[1] = 0xFFFFFFFF
    x[2] = np.uint32(1) << np.arange(32, dtype=np.uint32)
    d_in = _t(x, cuda_device)
    d_out = torch.zeros_like(d_in)
    hip.debug_stage(stage, np.float32, 1, None, 0, d_in, d_out, None, n)
    torch.cuda.synchronize()
    got = _np(d_out, np.uint32).reshape(n, 32)
    for i in range(0, n, 97):
        assert np.array_equal(got[i], oracle.transpose_bits(x[i])), i
    d_back = torch.zeros_like(d_in)
    hip.debug_stage(stage, np.float32, 1, None, 0, d_out, d_back, None, n)
    torch.cuda.synchronize()
    assert np.array_equal(_np(d_back, np.uint32).reshape(n, 32), x)


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
def test_forward_transform_matches_oracle(hiplib, cuda_device, profile, aligned):
    import torch

    from ndzip_amd import hip

    dtype, dims = profile
    wdt = word_dtype(dtype)
    data = _grid(dtype, dims, aligned, seed=3)
    nhc = oracle.num_hypercubes(data.shape)
    d_in = _t(data, cuda_device)
    for hc in sorted({0, 1, nhc // 2, nhc - 1}):
        d_out = torch.zeros(4096, dtype=d_in.dtype, device=cuda_device)
        hip.debug_stage(FWD, dtype, dims, data.shape, hc, d_in, d_out)
        torch.cuda.synchronize()
        cube = oracle.load_cube(data, hc)
        want = oracle.forward_transform(cube, dims)
        got = _np(d_out, wdt)
        assert np.array_equal(got, want), (hc, np.flatnonzero(got != want)[:8])


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("pattern", ["sparse", "random", "zeros", "ones", "single_bits", "dense_chunks"])
def test_residual_encoding_matches_oracle(hiplib, cuda_device, profile, pattern):
    import torch

    from ndzip_amd import hip

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
    d_in = _t(res, cuda_device)
    d_out = torch.zeros(4096 + 4096 // bits, dtype=d_in.dtype, device=cuda_device)
    d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    hip.debug_stage(ENC, dtype, dims, None, 0, d_in, d_out, d_len)
    torch.cuda.synchronize()
    n = int(d_len.cpu()[0])
    assert n == len(want)
    assert np.array_equal(_np(d_out, wdt)[:n], want)
    # and decode it back on the GPU (codec_profile_test.inl:735-801)
    d_res = torch.zeros(4096, dtype=d_in.dtype, device=cuda_device)
    d_stream = torch.zeros(4096 + 4096 // bits, dtype=d_in.dtype, device=cuda_device)
    d_stream[: len(want)] = _t(want, cuda_device)
    for stage in ([DEC, DEC_WIDE] if bits == 64 else [DEC]):
        d_res.zero_()
        hip.debug_stage(stage, dtype, dims, None, 0, d_stream, d_res)
        torch.cuda.synchronize()
        assert np.array_equal(_np(d_res, wdt), res), stage


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("aligned", [True, False], ids=["aligned", "unaligned"])
def test_inverse_transform_matches_oracle(hiplib, cuda_device, profile, aligned):
    import torch

    from ndzip_amd import hip

    dtype, dims = profile
    wdt = word_dtype(dtype)
    shape = _grid(dtype, dims, aligned, seed=0).shape
    nhc = oracle.num_hypercubes(shape)
    res = random_bits((4096,), dtype, 9).view(wdt)
    want_cube = oracle.inverse_transform(res, dims)
    n = int(np.prod(shape))
    for hc in sorted({0, nhc - 1}):
        for stage in ([INV, INV_WIDE] if wdt == np.uint64 else [INV]):
            d_out = torch.zeros(n, dtype=torch.int32 if wdt == np.uint32 else torch.int64, device=cuda_device)
            hip.debug_stage(stage, dtype, dims, shape, hc, _t(res, cuda_device), d_out)
            torch.cuda.synchronize()
            got = _np(d_out, wdt).reshape(shape)
            assert np.array_equal(oracle.load_cube(got.view(dtype), hc), want_cube), (hc, stage)
            # nothing outside the hypercube was touched
            assert np.count_nonzero(got) <= 4096


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_forward_then_inverse_is_identity(hiplib, cuda_device, profile):
    """block transform is reversible (codec_profile_test.inl:23-34), on the GPU stages."""
    import torch

    from ndzip_amd import hip

    dtype, dims = profile
    wdt = word_dtype(dtype)
    side = SIDE[dims]
    shape = (side,) * dims
    data = random_unit_floats(shape, dtype, 11)
    d_in = _t(data, cuda_device)
    d_res = torch.zeros(4096, dtype=d_in.dtype, device=cuda_device)
    d_back = torch.zeros(4096, dtype=d_in.dtype, device=cuda_device)
    hip.debug_stage(FWD, dtype, dims, shape, 0, d_in, d_res)
    hip.debug_stage(INV, dtype, dims, shape, 0, d_res, d_back)
    torch.cuda.synchronize()
    assert np.array_equal(_np(d_back, wdt), data.reshape(-1).view(wdt))


def test_wave_scan_and_sum(hiplib, cuda_device):
    """The DPP wave scan (row_shr 1/2/4/8 + row_bcast:15/31) and the wave sum (its last lane by v_readlane) on their own, against
    numpy: uint32 wraparound included."""
    import torch

    from ndzip_amd import hip

    rng = np.random.default_rng(4)
    n = 64 * 9
    x = rng.integers(0, 2**32, size=n, dtype=np.uint32)
    x[:64] = 1
    x[64:128] = 0xFFFFFFFF
    x[128:192] = np.arange(64, dtype=np.uint32)
    d_out = torch.zeros(n + n // 64, dtype=torch.int32, device=cuda_device)
    hip.debug_stage(6, np.float32, 1, None, 0, _t(x, cuda_device), d_out, None, n)
    torch.cuda.synchronize()
    got = _np(d_out, np.uint32)
    want = np.concatenate([np.cumsum(x.reshape(-1, 64).astype(np.uint64), axis=1).astype(np.uint32).reshape(-1),
                           x.reshape(-1, 64).astype(np.uint64).sum(axis=1).astype(np.uint32)])
    assert np.array_equal(got, want)


def _lookback_case(n, seed):
    """n tile lengths as the compress kernels publish them (128 .. 8448 words per two-hypercube tile), scaled down when n is
    large enough for their sum to leave 32 bits (the format's own limit: a stream has fewer than 2^32 words)."""
    rng = np.random.default_rng(seed)
    hi = min(8448, (2**32 - 1) // n)
    x = rng.integers(min(128, hi), hi + 1, size=n, dtype=np.uint32)
    x[: n // 7] = min(128, hi)  # a run of all-zero hypercubes: many equal small lengths in a row
    return x


def _check_lookback(cuda_device, n, grid, seed=5):
    import torch

    from ndzip_amd import hip

    x = _lookback_case(n, seed)
    d_out = torch.full((n + 2,), -1, dtype=torch.int32, device=cuda_device)
    hip.debug_stage(7, np.float32, 1, None, grid, _t(x, cuda_device), d_out, None, n)
    torch.cuda.synchronize()
    got = _np(d_out, np.uint32)
    incl = np.cumsum(x.astype(np.uint64))
    assert incl[-1] < 2**32
    assert got[n + 1] == 0, "look-back timeout"
    assert got[n] == incl[-1], "total"
    bad = np.flatnonzero(got[:n] != (incl - x).astype(np.uint32))
    assert bad.size == 0, f"first wrong prefixes at tiles {bad[:8]}"


@pytest.mark.parametrize("n,grid", [(1, 0), (63, 1), (4096, 1), (1 << 13, 16), (1 << 13, 17), (1 << 14, 0)])
def test_lookback_scan_on_its_own(hiplib, cuda_device, n, grid):
    """Stage 7: the production ticket / publish / look-back / release functions over n synthetic tile lengths against numpy's
    cumulative sum -- with one workgroup (every look-back finds its predecessor's inclusive prefix), sixteen (one ticket class),
    seventeen (sixteen classes) and the default grid; two launches on one scratch, so descriptor epochs and the ticket reset are
    part of it.  The reference pins its device-wide scan the same way (src/test/cuda_bits_test.cu:94-114)."""
    _check_lookback(cuda_device, n, grid)


@pytest.mark.hardware_only
@pytest.mark.parametrize("grid", [0, 16, 2048], ids=["default-grid", "16-workgroups", "2048-workgroups"])
def test_lookback_scan_at_a_million_tiles(hiplib, cuda_device, grid):
    """2^20 tiles -- twice the tile count of the largest configuration BASELINE names per GPU (16 GiB of float64: 524 288) and
    far beyond what any array in this suite reaches -- through the same functions."""
    _check_lookback(cuda_device, 1 << 20, grid, seed=6)
