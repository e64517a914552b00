"""Random plans through the C++ host of the multi-GPU path on the functional model (test tooling; CPU only):
`tools/fuzz_sharded_host.py <seed> <cases>` runs tests/cpp/sharded_threads.cc -- every rank a thread, the unchanged ndzip_amd/csrc/sharded.cc,
the kernels on tests/wavesim -- on `cases` random (extent, dtype, world, data mixture) and compares the stream the ranks wrote into one
buffer with the oracle's stream of the WHOLE array, byte for byte (the program itself checks both ways back: resident and loaded).
Extents include slabs without hypercubes, borders in every dimension, more ranks than hypercube planes, one rank.
Round 6: seeds 601-604 x 150 and 611-614 x 600 = 3 000 cases, all equal."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests import test_wavesim_fuzz as f  # noqa: E402
from tests.test_hip_sharded_native import THREADS_SRC  # noqa: E402
from tests.wavesim import build as simbuild  # noqa: E402


def build(workdir):
    lib = simbuild.build_sharded(variant="")
    here = os.path.dirname(lib)
    exe = os.path.join(workdir, "sharded_threads_model")
    subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", here, "-I", os.path.join(ROOT, "include"), THREADS_SRC, "-o", exe,
                    "-L" + here, "-l:" + os.path.basename(lib), "-l:libndzip_hip_wavesim.so", "-Wl,-rpath," + here], check=True)
    return exe


def main():
    seed, cases = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="fuzz_sharded")
    exe = build(work)
    env = dict(os.environ, WAVESIM_CUS="2", WAVESIM_BLOCKS_PER_CU="2")
    side = {1: 4096, 2: 64, 3: 16}
    for i in range(cases):
        dims = int(rng.integers(1, 4))
        dtype = [np.float32, np.float64][int(rng.integers(0, 2))]
        world = int(rng.integers(1, 9))
        s = side[dims]
        # dimension 0: 0 .. 9 hypercube planes (fewer planes than ranks happens), plus a tail; the others 0 .. 2 (1D: none) plus a tail
        extent = [int(rng.integers(0, 10)) * s + int(rng.integers(0, s if dims > 1 else 40))]
        for _ in range(1, dims):
            extent.append(int(rng.integers(0, 3)) * s + int(rng.integers(0, s)))
        extent = [max(1, e) for e in extent]
        if int(np.prod(extent)) > 400_000:
            extent[0] = max(1, extent[0] // 2)
        data = f._patterned(tuple(extent), dtype, int(rng.integers(0, 1 << 30)))
        data.tofile(os.path.join(work, "in.bin"))
        out = os.path.join(work, "out.bin")
        r = subprocess.run([exe, "--world", str(world), "--dtype", "f32" if dtype == np.float32 else "f64", "--extent", ",".join(map(str, extent)),
                            "--in", os.path.join(work, "in.bin"), "--out", out], capture_output=True, text=True, timeout=900, env=env)
        want = oracle.compress(data)
        ok = r.returncode == 0 and r.stdout.count(": ok") == world
        if ok:
            got = np.fromfile(out, dtype=want.dtype)
            ok = len(got) == len(want) and bool(np.array_equal(got, want))
        if not ok:
            print("FAILED", np.dtype(dtype).name, extent, world, r.returncode, r.stdout[-500:], r.stderr[-500:])
            sys.exit(1)
    print("ok", cases)


if __name__ == "__main__":
    main()
