"""Property test of the kernels on the functional model: for arbitrary small extents (with and without border, odd hypercube
counts, element-aligned or not), both value types and a mix of data patterns per hypercube (zeros, constants, ramps, noise with a
random number of low bits, raw random bits incl. NaN/Inf encodings, dense and sparse chunks next to each other), the stream equals
the oracle's bit for bit and decompress(compress(x)) == x.  (tests/wavesim is test infrastructure; see tests/test_wavesim_codec.py.)"""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import oracle
from tests.util import SIDE, same_bits, word_dtype
from tests.wavesim import sim


def _patterned(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    wdt = word_dtype(dtype)
    bits = np.dtype(wdt).itemsize * 8
    n = int(np.prod(shape))
    out = np.zeros(n, dtype=wdt)
    pos = 0
    while pos < n:  # runs of 1..6000 elements, each with its own pattern: hypercubes and chunks see mixtures
        run = int(rng.integers(1, 6000))
        kind = int(rng.integers(0, 7))
        m = min(run, n - pos)
        if kind == 0:
            seg = np.zeros(m, dtype=wdt)
        elif kind == 1:
            seg = np.full(m, rng.integers(0, np.iinfo(wdt).max, dtype=wdt, endpoint=True), dtype=wdt)
        elif kind == 2:
            seg = (np.arange(m, dtype=np.float64) * rng.random() * 1e-3).astype(dtype).view(wdt)
        elif kind == 3:
            seg = rng.integers(0, np.iinfo(wdt).max, size=m, dtype=wdt, endpoint=True)
        elif kind == 4:
            low = int(rng.integers(1, bits))
            base = rng.integers(0, np.iinfo(wdt).max, dtype=wdt, endpoint=True)
            seg = (base & ~wdt((1 << low) - 1)) | (rng.integers(0, np.iinfo(wdt).max, size=m, dtype=wdt, endpoint=True) & wdt((1 << low) - 1))
        elif kind == 5:
            seg = rng.random(m).astype(dtype).view(wdt)
        else:
            seg = np.where(rng.random(m) < 0.03, rng.integers(0, np.iinfo(wdt).max, size=m, dtype=wdt, endpoint=True), wdt(0)).astype(wdt)
        out[pos: pos + m] = seg
        pos += m
    return out.view(dtype).reshape(shape)


@st.composite
def _cases(draw):
    dtype = draw(st.sampled_from([np.float32, np.float64]))
    dims = draw(st.integers(1, 3))
    side = SIDE[dims]
    shape = []
    for d in range(dims):
        hcs = draw(st.sampled_from({1: [0, 1, 2, 3, 4, 5, 7], 2: [0, 1, 1, 2, 2, 3], 3: [0, 1, 1, 2, 2, 3]}[dims]))
        extra = draw(st.sampled_from([0, 0, 1, 3, side // 2, side - 1]))
        shape.append(max(1, hcs * side + extra) if (hcs or extra) else 1)
    return dtype, tuple(shape), draw(st.integers(0, 2 ** 31)), draw(st.sampled_from([(1, 1), (2, 2), (4, 3)])), draw(
        st.sampled_from(["", "reverse", "random:5"]))


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(_cases())
def test_model_equals_oracle_on_arbitrary_small_arrays(case):
    dtype, shape, seed, (cus, bpc), schedule = case
    data = _patterned(shape, dtype, seed)
    want = oracle.compress(data)
    got = sim.compress(data, cus=cus, blocks_per_cu=bpc, schedule=schedule)
    assert len(got) == len(want) and np.array_equal(got, want), (np.dtype(dtype).name, shape, seed)
    assert same_bits(sim.decompress(want, dtype, shape, schedule=schedule), data), (np.dtype(dtype).name, shape, seed)
