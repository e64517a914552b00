"""CPU model of the 256-work-item ("wide") encode mapping of ndzip_amd/csrc/codec_kernels_wide.hpp, checked against the oracle.

The HIP code itself is verified bit for bit on the GPU (tests/test_hip_*.py); this test pins the DESCRIPTION the kernel comments give
-- which lane ends up with which dword of which plane, which head bits it owns, where it writes -- by executing that description with
numpy for one hypercube of residuals and comparing with oracle.encode_cube.  If the mapping is ever changed, this is the executable spec."""
import numpy as np
import pytest

from oracle import oracle


def _transpose32(rows):
    """32 x uint32 -> 32 plane words, mirrored convention of the codec: plane i = bit 31-i of every row, row j -> bit 31-j."""
    rows = np.asarray(rows, dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint64)
    for i in range(32):
        bits = (rows >> np.uint64(31 - i)) & np.uint64(1)
        out[i] = sum(int(b) << (31 - j) for j, b in enumerate(bits))
    return out.astype(np.uint32)


def _encode_wide_f64(res):
    """res: 4096 uint64 residuals -> encoded run (uint64 words) exactly as the lane quads of compress_kernel_wide<u64> build it."""
    lanes = res.reshape(256, 16)
    heads = np.zeros(64, dtype=np.uint64)
    run32 = {}
    # chunk heads (OR over the quad) and positions (scan over chunks)
    for c in range(64):
        heads[c] = np.bitwise_or.reduce(lanes[4 * c: 4 * c + 4].reshape(-1))
    counts = np.array([bin(int(h)).count("1") for h in heads])
    excl = np.concatenate([[0], np.cumsum(counts)[:-1]])
    for t in range(256):
        c, q = t >> 2, t & 3
        pair = lanes[(t & ~1): (t & ~1) + 2]                       # the two lanes of the pair, 32 values
        vals = pair.reshape(-1)
        dwords = (vals >> np.uint64(32)) if (q & 1) == 0 else (vals & np.uint64(0xFFFFFFFF))   # even lane: HIGH dwords, odd: LOW
        planes = _transpose32(dwords)
        head_hi, head_lo = int(heads[c]) >> 32, int(heads[c]) & 0xFFFFFFFF
        head_bits = head_lo if (q & 1) else head_hi                 # lanes 1, 3: planes 32..63; lanes 0, 2: planes 0..31
        pos = 64 + int(excl[c]) + (bin(head_hi).count("1") if (q & 1) else 0)
        slot = 2 * pos + (0 if (q & 2) else 1)                      # lanes 0, 1 (values 0..31): HIGH dword of the plane word
        if q < 2:
            run32[2 * c + (1 if q == 0 else 0)] = head_hi if q == 0 else head_lo
        for i in range(32):
            if (head_bits >> (31 - i)) & 1:
                run32[slot] = int(planes[i])
                slot += 2
    n = 64 + int(counts.sum())
    words = np.zeros(2 * n, dtype=np.uint32)
    for k, v in run32.items():
        words[k] = v
    return words.view(np.uint64)


def _encode_wide_f32(res):
    lanes = res.reshape(256, 16)
    heads = np.array([np.bitwise_or.reduce(lanes[2 * c: 2 * c + 2].reshape(-1)) for c in range(128)], dtype=np.uint32)
    counts = np.array([bin(int(h)).count("1") for h in heads])
    excl = np.concatenate([[0], np.cumsum(counts)[:-1]])
    run = {}
    for t in range(256):
        c, odd = t >> 1, t & 1
        planes = _transpose32(lanes[2 * c: 2 * c + 2].reshape(-1))   # stage 16 crosses the pair, stages 8..1 stay in the lane
        mine = planes[16:] if odd else planes[:16]                    # even lane: planes 0..15, odd lane: planes 16..31
        head = int(heads[c])
        head_bits = ((head << 16) & 0xFFFFFFFF) if odd else head
        slot = 128 + int(excl[c]) + (bin(head >> 16).count("1") if odd else 0)
        if not odd:
            run[c] = head
        for i in range(16):
            if (head_bits >> (31 - i)) & 1:
                run[slot] = int(mine[i])
                slot += 1
    n = 128 + int(counts.sum())
    words = np.zeros(n, dtype=np.uint32)
    for k, v in run.items():
        words[k] = v
    return words


@pytest.mark.parametrize("kind", ["random", "sparse", "small"])
def test_wide_mapping_f64_matches_oracle(kind):
    rng = np.random.default_rng(11)
    res = rng.integers(0, 2 ** 63, 4096, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 4096, dtype=np.uint64)
    if kind == "sparse":
        res &= np.uint64(0x00FF00000F0F0001)
        res[rng.random(4096) < 0.7] = 0
    elif kind == "small":
        res &= np.uint64(0xFFFFF)
    assert np.array_equal(_encode_wide_f64(res), oracle.encode_cube(res))


@pytest.mark.parametrize("kind", ["random", "sparse", "small"])
def test_wide_mapping_f32_matches_oracle(kind):
    rng = np.random.default_rng(12)
    res = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
    if kind == "sparse":
        res &= np.uint32(0x0F0000F1)
        res[rng.random(4096) < 0.7] = 0
    elif kind == "small":
        res &= np.uint32(0x3FF)
    assert np.array_equal(_encode_wide_f32(res), oracle.encode_cube(res))
