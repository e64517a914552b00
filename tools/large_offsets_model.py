"""One array whose BYTE offsets cross 2^32 through the kernels on the functional model (test tooling; CPU only, ~20 GiB of RAM, minutes):
the single-GPU leg of the strong-scaling configuration (bench.py --config 16gib --gpus 1: one 16 GiB array on one GPU) is the only place
where a 32-bit byte offset inside a kernel would wrap, and the CPU suite's largest array is 1 GiB.  A 4.25 GiB grid -- 3D float64
544 x 1024 x 1024 or 3D float32 1088 x 1024 x 1024 -- is compressed and decompressed by the unchanged kernel sources (tests/wavesim) and
held against the oracle: stream bit for bit, round trip bit for bit.  The data differs from z-slab to z-slab, so a wrapped offset cannot
land on equal bytes.       usage: tools/large_offsets_model.py float64|float32|float32big [128|256: the 64-bit decoder mapping]
Round 6: both pass (see profiles/r06_code_object_rehearsal.txt)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndzip_amd.synth import synth_numpy  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.wavesim import sim  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "float64"
    # "float32big": 2064 x 1024 x 1024 float32 = 2^31 + 2^24 ELEMENTS (8.06 GiB; ~32 GiB of RAM): element offsets beyond a signed 32-bit int,
    # inside what index_type (uint32) and the C ABI allow for one array
    dtype = np.dtype("float32" if what == "float32big" else what)
    planes = 16
    slabs = 129 if what == "float32big" else 34 if dtype.itemsize == 8 else 68
    extent = (planes * slabs, 1024, 1024)
    wdt = np.uint64 if dtype.itemsize == 8 else np.uint32
    base = synth_numpy((planes, 1024, 1024), dtype.type, seed=1, noise_mask=0xFF).view(wdt)
    data = np.empty(extent, dtype=wdt)
    for k in range(slabs):  # every z-slab of hypercube planes its own low mantissa bits
        data[k * planes:(k + 1) * planes] = base ^ wdt(k * 37 + 1)
    data = data.view(dtype)
    print(f"{dtype.name} {'x'.join(map(str, extent))}: {data.nbytes / 2 ** 30:.2f} GiB, {data.nbytes // (4096 * dtype.itemsize)} hypercubes", flush=True)
    t = time.time()
    want = oracle.compress(data, num_threads=oracle.max_threads())
    print(f"oracle: {len(want)} words, ratio {want.nbytes / data.nbytes:.3f}, {time.time() - t:.0f} s", flush=True)
    t = time.time()
    got = sim.compress(data, cus=8, blocks_per_cu=4)
    print(f"model compress: {time.time() - t:.0f} s", flush=True)
    assert len(got) == len(want) and np.array_equal(got, want), "stream differs from the oracle"
    del got
    t = time.time()
    back = sim.decompress(want, dtype.type, extent, f64_work_items=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"model decompress: {time.time() - t:.0f} s", flush=True)
    assert np.array_equal(back.view(wdt), data.view(wdt)), "round trip differs"
    print("ok: stream == oracle, round trip bit-exact, byte offsets up to", hex(data.nbytes))


if __name__ == "__main__":
    main()
