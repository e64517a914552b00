#!/usr/bin/env python
"""Generate the golden fixtures from the REAL reference (oracle/_ref/libndzip_ref.so, compiled from /root/reference by
oracle/Makefile).  Runs only in the authoring container; the fixtures are data (inputs + expected streams / hashes).

  tests/golden/vectors.npz   small cases, byte-exact:  <name>__in, <name>__stream (+ shape/dtype in the name table)
  tests/golden/hashes.json   larger cases: shape, dtype, generator parameters, input sha256, stream sha256, words

usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndzip_amd.synth import synth_numpy  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SIDE = {1: 4096, 2: 64, 3: 16}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert oracle.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    small = {}
    table = []

    def add_small(name, data):
        data = np.ascontiguousarray(data)
        stream = oracle.ref_compress(data)
        back, consumed = oracle.ref_decompress(stream, data.dtype, data.shape)
        assert consumed == len(stream) and back.tobytes() == data.tobytes()
        small[name + "__in"] = data
        small[name + "__stream"] = stream
        table.append({"name": name, "shape": list(data.shape), "dtype": data.dtype.name, "words": int(len(stream))})

    rng = np.random.default_rng(20260928)
    for dt in (np.float32, np.float64):
        w = np.uint32 if dt == np.float32 else np.uint64
        for dims in (1, 2, 3):
            s = SIDE[dims]
            tag = f"{np.dtype(dt).name}_{dims}d"
            # one hypercube of uniform [0,1) values (reference tests: codec_profile_test.inl:952-995)
            add_small(f"unit_{tag}", rng.random((s,) * dims).astype(dt))
            # one hypercube of raw random bit patterns (NaN / Inf / denormal encodings included)
            add_small(f"bits_{tag}", rng.integers(0, np.iinfo(w).max, size=s ** dims, dtype=w, endpoint=True).view(dt).reshape((s,) * dims))
        # special values, 1D single hypercube
        add_small(f"zeros_{np.dtype(dt).name}", np.zeros(4096, dt))
        add_small(f"ones_{np.dtype(dt).name}", np.ones(4096, dt))
        z = np.zeros(4096, dt)
        z[0] = -0.0
        z[77] = np.inf
        z[78] = -np.inf
        z[79] = np.nan
        z[4095] = np.finfo(dt).tiny / 4
        add_small(f"special_{np.dtype(dt).name}", z)
        # zero hypercubes (codec_profile_test.inl:1045-1082) and tiny all-border arrays
        add_small(f"border_only_1d_{np.dtype(dt).name}", np.arange(5, dtype=dt))
        add_small(f"border_only_2d_{np.dtype(dt).name}", np.arange(63 * 5, dtype=dt).reshape(5, 63))
        add_small(f"border_only_3d_{np.dtype(dt).name}", np.arange(15 * 4 * 3, dtype=dt).reshape(3, 4, 15))
    # small bordered grids, odd hypercube counts (f64 header pad)
    add_small("bordered_1d_f32", synth_numpy((4096 + 3,), np.float32, 11, 0xFF))
    add_small("bordered_2d_f64_nhc3", synth_numpy((70, 200), np.float64, 12, 0xFF))   # NHC = 3 (odd) -> header pad
    add_small("bordered_3d_f32", synth_numpy((17, 35, 33), np.float32, 13, 0xFF))     # NHC = 4
    add_small("bordered_3d_f64_nhc1", synth_numpy((16, 17, 31), np.float64, 14, 0xFF))  # NHC = 1 (odd)
    add_small("three_cubes_1d_f64", np.zeros(3 * 4096, np.float64))
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **small)

    hashes = []
    rows = [((128, 128, 128), np.float32, 1, 0x0), ((128, 128, 128), np.float32, 1, 0xFF), ((128, 128, 128), np.float32, 1, 0xFFFF),
            ((1024, 1024), np.float64, 2, 0xFF), ((1 << 20,), np.float32, 3, 0xFF), ((64, 64, 64), np.float64, 4, 0xFF),
            ((50, 37, 41), np.float32, 5, 0xFF), ((200, 131), np.float64, 6, 0xFF), ((200, 70), np.float64, 6, 0xFF),
            ((12305,), np.float64, 7, 0xFF), ((255, 255), np.float32, 8, 0xFF), ((16383,), np.float32, 9, 0xFF),
            ((63, 63, 63), np.float64, 10, 0xFF), ((63, 63, 63), np.float32, 15, 0xFF), ((255, 255), np.float64, 16, 0xFF),
            ((16383,), np.float64, 17, 0xFF)]
    for shape, dt, seed, mask in rows:
        a = synth_numpy(shape, dt, seed, mask)
        s = oracle.ref_compress(a)
        hashes.append({"shape": list(shape), "dtype": np.dtype(dt).name, "seed": seed, "noise_mask": mask, "words": int(len(s)),
                       "input_sha256": sha(a), "stream_sha256": sha(s)})
    with open(os.path.join(HERE, "hashes.json"), "w") as f:
        json.dump({"generator": "ndzip_amd.synth.synth_numpy (SURVEY.md Appendix B)", "source": "reference serial CPU path, oracle/_ref",
                   "small_cases": table, "hashed_cases": hashes}, f, indent=1)
    print(f"{len(table)} small cases ({os.path.getsize(os.path.join(HERE, 'vectors.npz')) >> 10} KiB), {len(hashes)} hashed cases")


if __name__ == "__main__":
    main()
