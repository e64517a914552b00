"""End-to-end parity of the HIP path against the CPU oracle through the C ABI (bit-exact streams).

Mirrors src/test/codec_profile_test.inl:
  decode(encode(x)) == x for extent (4*side-1)^dims with the first chunk zeroed, all pairings   :37-96
  headers identical + same length                                                              :100-140
  single hypercube compresses identically / decompresses correctly                            :952-1043
  extents 0 and 1 (zero hypercubes)                                                            :1045-1082
plus the known answers recorded in SURVEY.md section 8a from the compiled reference, odd-NHC header padding
for f64, unaligned extents (scalar load path) and the host-pointer offloader."""
import numpy as np
import pytest

from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import (PROFILES, SIDE, device_compress, device_decompress, profile_id, random_bits, random_unit_floats,
                        same_bits, word_dtype)

pytestmark = pytest.mark.gpu


def _check_stream(data):
    want = oracle.compress(data)
    got = device_compress(data)
    assert len(got) == len(want), (len(got), len(want))
    nhc = oracle.num_hypercubes(data.shape)
    hdr = np.frombuffer(got.tobytes(), dtype=np.uint32)[:nhc]
    hdr_want = np.frombuffer(want.tobytes(), dtype=np.uint32)[:nhc]
    assert np.array_equal(hdr, hdr_want), "header entries differ"
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"first differing words {bad[:8]}"
    return got


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_roundtrip_all_pairings_with_border(hiplib, cuda_device, profile):
    dtype, dims = profile
    n = SIDE[dims] * 4 - 1
    if dims == 1:
        n = SIDE[dims] * 4 - 1
    shape = (n,) * dims
    data = random_unit_floats(shape, dtype, 21)
    data.reshape(-1)[: np.dtype(dtype).itemsize * 8] = 0  # regression input of :49-50
    stream = _check_stream(data)
    # HIP compress => HIP decompress, HIP compress => oracle decompress, oracle compress => HIP decompress
    assert same_bits(device_decompress(stream, dtype, shape), data)
    back, consumed = oracle.decompress(stream, dtype, shape)
    assert consumed == len(stream) and same_bits(back, data)
    assert same_bits(device_decompress(oracle.compress(data), dtype, shape), data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_single_hypercube(hiplib, cuda_device, profile):
    dtype, dims = profile
    shape = (SIDE[dims],) * dims
    data = random_unit_floats(shape, dtype, 22)
    stream = _check_stream(data)
    assert same_bits(device_decompress(stream, dtype, shape), data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("n", [0, 1])
def test_zero_hypercubes(hiplib, cuda_device, profile, n):
    dtype, dims = profile
    shape = (n,) * dims
    data = np.full(shape, 42, dtype=dtype)
    stream = _check_stream(data)
    assert len(stream) == n
    if n:
        assert same_bits(device_decompress(stream, dtype, shape), data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("kind", ["bits", "unit", "synthetic"])
def test_many_hypercubes_unaligned_and_aligned(hiplib, cuda_device, profile, kind):
    from ndzip_amd.synth import synth_numpy

    dtype, dims = profile
    side = SIDE[dims]
    shapes = {1: [(side * 37,), (side * 5 + 17,)], 2: [(side * 5, side * 6), (side * 3 + 1, side * 4 + 3)],
              3: [(side * 3, side * 4, side * 6), (side * 2 + 5, side * 3 + 1, side * 3 + 2)]}[dims]
    for i, shape in enumerate(shapes):
        if kind == "bits":
            data = random_bits(shape, dtype, 30 + i)
        elif kind == "unit":
            data = random_unit_floats(shape, dtype, 40 + i)
        else:
            data = synth_numpy(shape, dtype, seed=50 + i, noise_mask=0xFF)
        stream = _check_stream(data)
        assert same_bits(device_decompress(stream, dtype, shape), data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_dense_and_sparse_chunks_mixed(hiplib, cuda_device, profile):
    """Chunks that keep ALL their planes take a 16-byte-access path in the encoder (f32) and the decoder (f32, f64) when
    they start on a 16-byte boundary of the run; whether they do depends on how many planes the chunks in front of them
    kept.  Random bits with constant spans, spans with the low bits cleared and single zeros strewn in: wavefronts in which
    dense-aligned, dense-unaligned and ordinary chunks sit side by side, in every order."""
    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 6,), 2: (side * 3, side * 2), 3: (side * 2, side, side * 3)}[dims]
    w = word_dtype(dtype)
    bits = 8 * np.dtype(dtype).itemsize
    rng = np.random.default_rng(4711 + dims)
    words = random_bits(shape, dtype, 90 + dims).view(w).reshape(-1).copy()
    n = words.size
    for _ in range(n // 400):  # constant spans: all-zero residuals, chunks that keep few planes
        a = int(rng.integers(0, n - 200))
        words[a:a + int(rng.integers(1, 200))] = words[a]
    for _ in range(n // 400):  # spans without their low k bits: chunks that keep bits - k planes or so
        a = int(rng.integers(0, n - 200))
        k = int(rng.integers(1, bits - 1))
        words[a:a + int(rng.integers(1, 200))] &= w(np.iinfo(w).max >> k << k)
    data = words.view(dtype).reshape(shape)
    stream = _check_stream(data)
    assert same_bits(device_decompress(stream, dtype, shape), data)
    assert same_bits(device_decompress(oracle.compress(data), dtype, shape), data)


@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("work_items", [128, 256])
def test_f64_decoder_mappings_agree(hiplib, cuda_device, dims, work_items):
    """ndzip_hip_decompressor_set_f64_work_items: the 64-bit decoder with 256 work-items per hypercube (decompress_kernel_wide) and with 128
    (decompress_kernel; every other float64 test in this suite runs both, tests/util.py) decode the oracle's streams to the
    same bits -- incompressible hypercubes (the wave-uniform dense path of the wide decoder: every chunk of a wavefront keeps all
    64 planes), smooth data, zeros, the mixed dense / sparse pattern, unaligned rows with a border."""
    import ndzip_amd

    dtype = np.float64
    side = SIDE[dims]
    shapes = {1: [(side * 5,), (side * 3 + 17,)], 2: [(side * 2, side * 3), (side * 2 + 3, side + 9)],
              3: [(side * 2, side, side * 2), (side + 1, side * 2 + 2, side + 3)]}[dims]
    for i, shape in enumerate(shapes):
        rng = np.random.default_rng(100 * dims + i)
        mixed = random_bits(shape, dtype, 70 + i).view(np.uint64).reshape(-1).copy()
        for _ in range(max(1, mixed.size // 500)):
            a = int(rng.integers(0, max(1, mixed.size - 300)))
            mixed[a:a + int(rng.integers(1, 300))] &= np.uint64(0xFFFFFFFFFFFFFFFF >> int(rng.integers(1, 63)))
        for data in (random_bits(shape, dtype, 60 + i), random_unit_floats(shape, dtype, 61 + i), np.zeros(shape, dtype),
                     synth_numpy(shape, dtype, seed=62 + i, noise_mask=0xFFFF), mixed.view(dtype).reshape(shape)):
            stream = oracle.compress(data)
            assert same_bits(device_decompress(stream, dtype, shape, f64_work_items=work_items), data)
    dec = ndzip_amd.make_hip_decompressor(dtype, dims)
    with pytest.raises(ndzip_amd.NdzipHipError):
        dec.set_f64_work_items(64)
    dec.close()


def test_zz_both_f64_decoder_kernels_agreed(hiplib, cuda_device):
    """Every float64 test that did not choose a decoder mapping decoded with BOTH 64-bit kernels (tests/util.py::device_decompress) and
    recorded disagreements instead of failing on the spot; this test -- sorted behind the 64-bit tests by tests/conftest.py -- is where
    they surface.  (test_f64_decoder_mappings_agree checks each kernel against the ORACLE on its own.)"""
    from tests import util

    assert util.F64_DECODER_DISAGREEMENTS == [], util.F64_DECODER_DISAGREEMENTS[:10]


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
@pytest.mark.parametrize("skew", [1, 3])
def test_element_aligned_device_pointers(hiplib, cuda_device, profile, skew):
    """Array, stream and output pointers that are only element-aligned (sub-views `skew` elements into an allocation):
    the 16-byte global accesses of the kernels must not assume more (the reference takes any T* / bits_type*)."""
    import torch

    import ndzip_amd
    from ndzip_amd.synth import synth_numpy

    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 3 + 5,), 2: (side * 2 + 3, side * 3), 3: (side * 2, side + 3, side * 2 + 1)}[dims]
    data = synth_numpy(shape, dtype, seed=77 + skew, noise_mask=0xFF)
    expect = oracle.compress(data)
    n = data.size
    wdt = torch.int32 if np.dtype(dtype) == np.float32 else torch.int64
    ndt = np.int32 if np.dtype(dtype) == np.float32 else np.int64
    bound = ndzip_amd.compressed_length_bound(dtype, shape)
    d_in_alloc = torch.zeros(n + 8, dtype=wdt, device=cuda_device)
    d_in = d_in_alloc[skew: skew + n]
    d_in.copy_(torch.from_numpy(data.reshape(-1).view(ndt)))
    d_stream = torch.zeros(bound + 8, dtype=wdt, device=cuda_device)[skew: skew + bound]
    d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    stream = torch.cuda.current_stream().cuda_stream
    comp = ndzip_amd.make_hip_compressor(dtype, ndzip_amd.CompressorRequirements(shape), stream)
    comp.compress(d_in, shape, d_stream, d_len)
    comp.check()
    words = int(d_len.cpu().numpy().view(np.uint32)[0])
    got = d_stream[:words].cpu().numpy().view(expect.dtype)
    assert words == len(expect) and np.array_equal(got, expect)
    d_out = torch.zeros(n + 8, dtype=wdt, device=cuda_device)[skew: skew + n]
    dec = ndzip_amd.make_hip_decompressor(dtype, dims, stream)
    dec.decompress(d_stream, d_out, shape)
    dec.check()
    assert same_bits(d_out.cpu().numpy().view(dtype).reshape(shape), data)
    comp.close()
    dec.close()


def test_f64_odd_hypercube_count_zeroes_header_pad(hiplib, cuda_device):
    # 1D f64 3x4096 zeros -> len 194, header {0x40, 0x80, 0xc0, pad 0}
    data = np.zeros(3 * 4096, dtype=np.float64)
    stream = _check_stream(data)
    assert len(stream) == 194
    assert stream[0] == 0x0000008000000040 and stream[1] == 0x00000000000000C0
    data2 = random_unit_floats((200, 70), np.float64, 7)  # NHC = 3
    stream2 = _check_stream(data2)
    assert np.frombuffer(stream2.tobytes(), dtype=np.uint32)[3] == 0


def test_known_answers_from_reference(hiplib, cuda_device):
    """Known answers captured from the compiled reference."""
    s = device_compress(np.zeros(4096, np.float32))
    assert len(s) == 129 and s[0] == 0x80 and not s[1:].any()
    s = device_compress(np.ones(4096, np.float32))
    assert len(s) == 136 and s[0] == 0x87 and s[1] == 0x7F000000 and not s[2:129].any() and (s[129:] == 0x80000000).all()
    s = device_compress(np.ones((16, 16, 16), np.float32))
    assert len(s) == 136 and s[0] == 0x87
    s = device_compress(np.ones(4096, np.float64))
    assert len(s) == 75 and s[0] == 0x4A and s[1] == 0x7FE0000000000000 and (s[65:] == 0x8000000000000000).all()
    z = np.zeros(4096, np.float32)
    z[0] = -0.0
    s = device_compress(z)
    assert len(s) == 131 and s[1] == 0x80000001 and s[129] == 0x40000000 and s[130] == 0x80000000
    b = np.zeros(4099, np.float32)
    b[4096:] = [1, 2, -1]
    s = device_compress(b)
    assert len(s) == 132 and list(s[-3:]) == [0x3F800000, 0x40000000, 0xBF800000]
    s = device_compress(np.arange(5, dtype=np.float32))
    assert len(s) == 5


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_host_pointer_offloader(hiplib, cuda_device, profile):
    """offloader<T>::compress / decompress semantics (offload.hh:16-24): return values and kernel_duration."""
    import ndzip_amd

    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 3 + 9,), 2: (side * 2 + 5, side * 3), 3: (side * 2, side * 2 + 3, side * 2)}[dims]
    data = random_unit_floats(shape, dtype, 60)
    off = ndzip_amd.make_hip_offloader(dtype, dims)
    stream = off.compress(data)
    assert off.last_kernel_ns > 0
    assert np.array_equal(stream, oracle.compress(data))
    back, consumed = off.decompress(stream, shape)
    assert consumed == len(stream)
    assert same_bits(back, data)
    with pytest.raises(ndzip_amd.NdzipHipError, match="dimensionality"):
        ndzip_amd.make_hip_offloader(dtype, 3 if dims != 3 else 2).compress(data)


def test_compressor_reuse_and_capacity(hiplib, cuda_device):
    """One compressor object serves several extents up to its requirements (cuda_codec.inl:543-552)."""
    import torch

    import ndzip_amd

    req = ndzip_amd.CompressorRequirements((64 * 3, 64 * 2), (64, 64 * 5))
    assert req.max_num_hypercubes == 6
    comp = ndzip_amd.make_hip_compressor(np.float32, req)
    for shape in [(64 * 3, 64 * 2), (64, 64 * 5), (70, 130)]:
        data = random_unit_floats(shape, np.float32, 70)
        d_in = torch.from_numpy(data).to(cuda_device)
        d_out = torch.zeros(ndzip_amd.compressed_length_bound(np.float32, shape), dtype=torch.int32, device=cuda_device)
        d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        comp.compress(d_in, shape, d_out, d_len)
        comp.check()
        n = int(d_len.cpu()[0])
        assert np.array_equal(d_out[:n].cpu().numpy().view(np.uint32), oracle.compress(data))
    with pytest.raises(ndzip_amd.NdzipHipError):
        comp.compress(d_in, (64 * 4, 64 * 4), d_out, d_len)
    with pytest.raises(ndzip_amd.NdzipHipError, match="dimensionality"):
        comp.compress(d_in, (4096,), d_out, d_len)


@pytest.mark.parametrize("dtype,shape,limit", [(np.float32, (16 * 7 + 3, 32, 16), 16 * 2 * 32 * 16 + 100), (np.float64, (64 * 5, 64 * 2), 64 * 64 * 2 + 1),
                                              (np.float32, (4096 * 6 + 77,), 4096 * 2 + 5)])
def test_chunked_interface_for_arrays_beyond_the_format_limits(hiplib, cuda_device, dtype, shape, limit):
    """ndzip_hip_chunked_*: one independent stream per slab of dimension 0, concatenated (the reference tool's multi-array
    file format, compress.cc:34-45), with the element limit lowered so that small arrays need several slabs."""
    from ndzip_amd import hip
    from ndzip_amd.synth import synth_numpy

    data = synth_numpy(shape, dtype, seed=21, noise_mask=0xFFF)
    rows, n, bound = hip.chunked_plan(dtype, shape, limit)
    assert n > 1
    want = np.concatenate([oracle.compress(data[k * rows: (k + 1) * rows]) for k in range(n)])
    got = hip.chunked_compress(data, limit)
    assert len(got) == len(want) and np.array_equal(got, want)
    back, consumed = hip.chunked_decompress(want, dtype, shape, limit)
    assert consumed == len(want) and same_bits(back, data)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_corrupt_header_entries_decode_as_zeros_not_as_stale_memory(hiplib, cuda_device, profile):
    """include/ndzip_hip.h: a header entry the format rules out makes the affected hypercubes (and, for the last entry, the
    border) decode as ZEROS with the error word set.  The output buffer is pre-filled with a canary pattern: nothing of it may
    survive, and the untouched hypercubes still decode bit-exactly.  (The reference trusts the header: cuda_codec.inl:628-652.)"""
    import torch

    import ndzip_amd

    dtype, dims = profile
    side = SIDE[dims]
    shape = {1: (side * 4 + 9,), 2: (side * 2 + 3, side * 2), 3: (side, side * 2 + 1, side * 2)}[dims]
    data = random_unit_floats(shape, dtype, 19)
    nhc = oracle.num_hypercubes(shape)
    assert nhc == 4
    wdt = torch.int32 if np.dtype(dtype) == np.float32 else torch.int64
    canary = 0x5A5A5A5A if np.dtype(dtype) == np.float32 else 0x5A5A5A5A5A5A5A5A
    n = int(np.prod(shape))
    # per element: either the original value or exactly zero
    for victim in (1, nhc - 1):
        stream = oracle.compress(data).copy()
        header = stream.view(np.uint32)
        header[victim] = 0xFFFFFF00 if victim != nhc - 1 else header[victim] + 5 * (4096 + 128)
        d_stream = torch.from_numpy(stream.view(np.int32 if stream.dtype == np.uint32 else np.int64)).to(cuda_device)
        d_out = torch.full((n,), canary, dtype=wdt, device=cuda_device)
        dec = ndzip_amd.make_hip_decompressor(dtype, dims, torch.cuda.current_stream().cuda_stream)
        dec.decompress(d_stream, d_out, shape, stream_length_words=len(stream))
        with pytest.raises(ndzip_amd.NdzipHipError, match="corrupt stream header"):
            dec.check()
        dec.close()
        got = d_out.cpu().numpy().view(word_dtype(dtype)).reshape(shape)
        want = np.ascontiguousarray(data).view(word_dtype(dtype))
        assert not (got == canary).any(), "part of the output was left unwritten"
        differs = got != want
        assert (got[differs] == 0).all(), "a rejected region holds something other than zeros"
        assert differs.any() and differs.sum() <= 2 * 4096 + oracle_border_elements(shape)
        # a corrupt entry takes out its own hypercube and the one whose start it is; the first hypercube is never affected here
        first = tuple(slice(0, side) for _ in range(dims))
        assert not differs[first].any()


def oracle_border_elements(shape):
    side = SIDE[len(shape)]
    inner = 1
    for e in shape:
        inner *= e // side * side
    return int(np.prod(shape)) - inner


def test_workgroups_per_cu_cap_changes_the_grid_not_the_stream(hiplib, cuda_device):
    """ndzip_hip_compressor_set_max_workgroups_per_cu: 1, 2 and 3 workgroups per CU (the persistent grid of rounds 1-2 was 3) and the
    default (4) produce the same stream on one handle; the cap is a tuning / diagnosis knob without a reference counterpart."""
    import torch

    import ndzip_amd

    shape = (48, 64, 96)
    data = random_unit_floats(shape, np.float32, 5)
    want = oracle.compress(data)
    d_in = torch.from_numpy(data).to(cuda_device)
    d_out = torch.zeros(ndzip_amd.compressed_length_bound(np.float32, shape), dtype=torch.int32, device=cuda_device)
    d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    comp = ndzip_amd.make_hip_compressor(np.float32, ndzip_amd.CompressorRequirements(shape), torch.cuda.current_stream().cuda_stream)
    for cap in (3, 1, 0, 2):
        comp.set_max_workgroups_per_cu(cap)
        d_out.zero_()
        comp.compress(d_in, shape, d_out, d_len)
        comp.check()
        n = int(d_len.cpu()[0])
        assert np.array_equal(d_out[:n].cpu().numpy().view(np.uint32), want), cap
    with pytest.raises(ndzip_amd.NdzipHipError):
        comp.set_max_workgroups_per_cu(-1)
    comp.close()
