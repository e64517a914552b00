"""CPU suite (-m "not gpu"): pins the oracle -- the plain-C restatement under oracle/ -- against
  (a) byte-exact streams produced by the REAL reference (tests/golden/, generator make_golden.py),
  (b) the known answers the reference's own tests hold (for_each_border_slice, src/test/codec_generic_test.cc:102-111),
  (c) the known answers recorded from the compiled reference in SURVEY.md section 8a / Appendix B,
  (d) the reference library itself when oracle/_ref/libndzip_ref.so is present (authoring container),
and restates the reference's generic tests (codec_generic_test.cc, codec_profile_test.inl:23-34)."""
import hashlib
import json
import os

import numpy as np
import pytest

from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import PROFILES, SIDE, profile_id, random_bits, random_unit_floats, same_bits, sparse_residuals, word_dtype

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _golden():
    with open(os.path.join(GOLDEN, "hashes.json")) as f:
        meta = json.load(f)
    vec = np.load(os.path.join(GOLDEN, "vectors.npz"))
    return meta, vec


META, VEC = _golden()


@pytest.mark.parametrize("case", META["small_cases"], ids=lambda c: c["name"])
def test_golden_small_vectors(case):
    data = VEC[case["name"] + "__in"]
    want = VEC[case["name"] + "__stream"]
    assert list(data.shape) == case["shape"] and data.dtype.name == case["dtype"]
    got = oracle.compress(data)
    assert len(got) == case["words"]
    assert np.array_equal(got, want)
    assert np.array_equal(oracle.compress(data, num_threads=4), want)
    back, consumed = oracle.decompress(want, data.dtype, data.shape)
    assert consumed == len(want)
    assert same_bits(back, data)


@pytest.mark.parametrize("case", META["hashed_cases"], ids=lambda c: f"{c['dtype']}-{'x'.join(map(str, c['shape']))}-{c['noise_mask']:#x}")
def test_golden_hashed_streams(case):
    data = synth_numpy(case["shape"], np.dtype(case["dtype"]).type, case["seed"], case["noise_mask"])
    assert hashlib.sha256(data.tobytes()).hexdigest() == case["input_sha256"], "generator drifted"
    stream = oracle.compress(data, num_threads=2)
    assert len(stream) == case["words"]
    assert hashlib.sha256(stream.tobytes()).hexdigest() == case["stream_sha256"]
    back, consumed = oracle.decompress(stream, data.dtype, data.shape, num_threads=2)
    assert consumed == len(stream) and same_bits(back, data)


def test_appendix_b_rows():
    """Rows of SURVEY.md Appendix B (captured from the compiled reference by the survey, independent of make_golden)."""
    rows = [((128, 128, 128), np.float32, 1, 0x0, 1288946, "dd5964f087c62b9f", "9f0a155a11bf4138"),
            ((1024, 1024), np.float64, 2, 0xFF, 835819, "cc0b72c1137e716d", "f3d63aa82d671435"),
            ((200, 70), np.float64, 6, 0xFF, 11190, "91548a01ec3e3014", "5b5a164e8c12aa1c"),
            ((63, 63, 63), np.float64, 10, 0xFF, 224640, "9f8df03e809a6bbb", "901c323ff32baf42")]
    for shape, dt, seed, mask, words, ih, sh in rows:
        a = synth_numpy(shape, dt, seed, mask)
        assert hashlib.sha256(a.tobytes()).hexdigest().startswith(ih)
        s = oracle.compress(a)
        assert len(s) == words and hashlib.sha256(s.tobytes()).hexdigest().startswith(sh)


def test_border_slices_known_answers():
    """src/test/codec_generic_test.cc:102-111"""
    B = oracle.border_slices
    assert B((4, 4), 4) == []
    assert B((4, 6), 2) == []
    assert B((5, 4), 4) == [(16, 4)]
    assert B((4, 5), 4) == [(4, 1), (9, 1), (14, 1), (19, 1)]
    assert B((4, 5), 2) == [(4, 1), (9, 1), (14, 1), (19, 1)]
    assert B((4, 6), 4) == [(4, 2), (10, 2), (16, 2), (22, 2)]
    assert B((4, 6), 5) == [(0, 24)]
    assert B((6, 4), 5) == [(0, 24)]
    # 3D: one slice per recursion level, increasing linear index
    assert B((5, 5, 5), 4) == [(4, 1), (9, 1), (14, 1), (19, 1), (20, 5), (29, 1), (34, 1), (39, 1), (44, 1), (45, 5),
                              (54, 1), (59, 1), (64, 1), (69, 1), (70, 5), (79, 1), (84, 1), (89, 1), (94, 1), (95, 5), (100, 25)]


def test_known_answers_from_reference():
    """SURVEY.md section 8a"""
    s = oracle.compress(np.zeros(4096, np.float32))
    assert len(s) == 129 and s[0] == 0x80 and not s[1:].any()
    s = oracle.compress(np.ones(4096, np.float32))
    assert len(s) == 136 and s[0] == 0x87 and s[1] == 0x7F000000 and (s[129:] == 0x80000000).all()
    s = oracle.compress(np.ones(4096, np.float64))
    assert len(s) == 75 and s[0] == 0x4A and s[1] == 0x7FE0000000000000 and (s[65:] == 0x8000000000000000).all()
    z = np.zeros(4096, np.float32)
    z[0] = -0.0
    s = oracle.compress(z)
    assert len(s) == 131 and s[1] == 0x80000001 and s[129] == 0x40000000 and s[130] == 0x80000000
    b = np.zeros(4099, np.float32)
    b[4096:] = [1, 2, -1]
    s = oracle.compress(b)
    assert len(s) == 132 and list(s[-3:]) == [0x3F800000, 0x40000000, 0xBF800000]
    s = oracle.compress(np.zeros(3 * 4096, np.float64))
    assert len(s) == 194 and s[0] == 0x0000008000000040 and s[1] == 0x00000000000000C0
    assert len(oracle.compress(np.arange(5, dtype=np.float32))) == 5


@pytest.mark.parametrize("wdt", [np.uint32, np.uint64])
def test_bit_transpose_is_involution_and_matches_definition(wdt):
    """codec_generic_test.cc:65-81, plus fast network == transpose_bits_trivial (cpu_codec.inl:355-363)"""
    import ctypes as C

    bits = np.dtype(wdt).itemsize * 8
    rng = np.random.default_rng(1)
    L = oracle.lib()
    for _ in range(50):
        x = rng.integers(0, np.iinfo(wdt).max, size=bits, dtype=wdt, endpoint=True) >> wdt(rng.integers(0, bits))
        t = oracle.transpose_bits(x)
        ref = np.zeros_like(x)
        getattr(L, f"ndzip_oracle_transpose_bits_trivial_u{bits}")(C.c_void_p(x.ctypes.data), C.c_void_p(ref.ctypes.data))
        assert np.array_equal(t, ref)
        assert np.array_equal(oracle.transpose_bits(t), x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_zero_word_compaction_is_reversible(dtype):
    """codec_generic_test.cc:38-62 (head / zero-map convention) at the hypercube level"""
    for res in (sparse_residuals(dtype, 3), random_bits((4096,), dtype, 4).view(word_dtype(dtype)), np.zeros(4096, word_dtype(dtype))):
        enc = oracle.encode_cube(res)
        bits = res.dtype.itemsize * 8
        heads = enc[: 4096 // bits]
        assert len(enc) == 4096 // bits + sum(bin(int(h)).count("1") for h in heads)
        dec, consumed = oracle.decode_cube(enc)
        assert consumed == len(enc) and np.array_equal(dec, res)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_block_transform_is_reversible(profile):
    """codec_profile_test.inl:23-34"""
    dtype, dims = profile
    x = random_bits((4096,), dtype, 5).view(word_dtype(dtype))
    assert np.array_equal(oracle.inverse_transform(oracle.forward_transform(x, dims), dims), x)


@pytest.mark.parametrize("dims", [1, 2, 3])
def test_hypercube_layout_is_sane(dims):
    """codec_generic_test.cc:114-157: every element inside a full hypercube is visited exactly once, in row-major
    hypercube order."""
    n = {1: 3 * 4096 + 7, 2: 100, 3: 100}[dims]
    shape = (n,) * dims
    side = SIDE[dims]
    idx = np.arange(n ** dims, dtype=np.float64).reshape(shape)
    nhc = oracle.num_hypercubes(shape)
    assert nhc == (n // side) ** dims
    seen = np.zeros(n ** dims, dtype=bool)
    prev_origin = -1
    for hc in range(nhc):
        cube = oracle.load_cube(idx, hc).view(np.float64).astype(np.int64)
        assert not seen[cube].any()
        seen[cube] = True
        origin = np.unravel_index(cube[0], shape)
        assert all(o % side == 0 for o in origin)
        lin = np.ravel_multi_index(tuple(o // side for o in origin), (n // side,) * dims)
        assert lin == hc and lin > prev_origin
        prev_origin = lin
    assert seen.sum() == nhc * 4096 == n ** dims - oracle.border_count(shape)


@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_roundtrip_with_border_and_thread_counts(profile):
    """codec_profile_test.inl:37-96 (serial <-> multi-threaded pairings), :100-140 (headers identical)"""
    dtype, dims = profile
    n = SIDE[dims] * 4 - 1 if dims > 1 else SIDE[dims] * 4 - 1
    shape = (n,) * dims if dims < 3 else (SIDE[3] * 4 - 1,) * 3
    data = random_unit_floats(shape, dtype, 8)
    data.reshape(-1)[: np.dtype(dtype).itemsize * 8] = 0
    s1 = oracle.compress(data, 1)
    s4 = oracle.compress(data, 4)
    assert np.array_equal(s1, s4)
    for t_dec in (1, 3):
        back, consumed = oracle.decompress(s1, dtype, shape, t_dec)
        assert consumed == len(s1) and same_bits(back, data)


def test_length_bound_formula():
    """compressed_length_bound, src/ndzip/common.cc:31-55"""
    for dtype, bits in ((np.float32, 32), (np.float64, 64)):
        for shape in [(0,), (1,), (4096,), (4099,), (64, 64), (70, 200), (16, 16, 16), (17, 35, 33), (5, 4, 3)]:
            nhc = oracle.num_hypercubes(shape)
            want = -(-nhc // (bits // 32)) + nhc * (4096 // bits) * (bits + 1) + oracle.border_count(shape)
            assert oracle.compressed_length_bound(dtype, shape) == want


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (only in the authoring container)")
@pytest.mark.parametrize("profile", PROFILES, ids=profile_id)
def test_oracle_equals_reference_library(profile):
    dtype, dims = profile
    side = SIDE[dims]
    shapes = {1: [(side * 2 + 5,), (17,)], 2: [(side + 3, side * 2 + 1), (side * 2, side)], 3: [(side + 1, side * 2, side + 5), (side, side, side * 2)]}[dims]
    for i, shape in enumerate(shapes):
        for data in (random_unit_floats(shape, dtype, 100 + i), random_bits(shape, dtype, 200 + i)):
            want = oracle.ref_compress(data)
            assert np.array_equal(oracle.compress(data), want)
            back, consumed = oracle.ref_decompress(oracle.compress(data, 2), dtype, shape)
            assert consumed == len(want) and same_bits(back, data)
            assert oracle.compressed_length_bound(dtype, shape) == oracle.ref_compressed_length_bound(dtype, shape)
