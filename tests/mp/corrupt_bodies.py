"""Corrupt BODIES behind a valid header through the decoders (run by tests/test_wavesim_asan.py on the AddressSanitizer build of the functional
model): chunk heads that claim every plane, random words, all ones, single bit flips -- any bits may come out, but no access may leave a buffer:
a run is fetched by its header's bounds and everything behind that is LDS-local.  usage: corrupt_bodies.py <seed> <trials per profile>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ndzip_amd import hip
from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.wavesim import sim

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = 0
for dtype, extent in ((np.float32, (4096 * 3 + 9,)), (np.float32, (70, 130)), (np.float32, (33, 18, 35)), (np.float64, (4096 * 2 + 5,)), (np.float64, (130, 70)), (np.float64, (17, 34, 33))):
    data = synth_numpy(extent, dtype, seed=3, noise_mask=0xFF)
    good = oracle.compress(data)
    nhc = hip.num_hypercubes(extent)
    hw = hip.header_words(dtype, nhc)
    bits = good.dtype.itemsize * 8
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        s = good.copy()
        body = s[hw:]
        kind = trial % 4
        if kind == 0:      # every chunk head claims all planes
            body[rng.integers(0, len(body), size=max(1, len(body) // 50))] = np.iinfo(good.dtype).max
        elif kind == 1:    # random words
            idx = rng.integers(0, len(body), size=max(1, len(body) // 10))
            body[idx] = rng.integers(0, np.iinfo(good.dtype).max, size=len(idx), dtype=good.dtype, endpoint=True)
        elif kind == 2:    # all ones everywhere
            body[:] = np.iinfo(good.dtype).max
        else:              # single bit flips
            idx = rng.integers(0, len(body), size=64)
            body[idx] ^= (good.dtype.type(1) << rng.integers(0, bits, size=64).astype(good.dtype))
        for bounded in (False, True):
            for wi in ((0,) if dtype == np.float32 else (128, 256)):
                out = sim.decompress(s, dtype, extent, bounded=bounded, f64_work_items=wi)   # any bits may come out; no access may leave a buffer
                assert out.shape == tuple(extent)
                cases += 1
print("ok", cases)
