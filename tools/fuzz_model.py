"""Extended property test of the kernels on the functional model (test tooling; CPU only): `tools/fuzz_model.py <seed> <cases>` runs
`cases` random extents / value types / data mixtures / model schedules (the generators of tests/test_wavesim_fuzz.py, NOT derandomised)
against the oracle: streams bit for bit, decompress(compress(x)) == x.  Build the model first (python -c "from tests.wavesim import build
as b; b.build()") when running several seeds side by side.  Round 3: seeds 11-14 x 400, 21-24 x 300, 31-34 x 2500 -- 12 800 cases, all equal; and seeds 41-44 x 700 on the
AddressSanitizer build of the model (LD_PRELOAD of the sanitizer runtime, WAVESIM_VARIANT=asan as in tests/test_wavesim_asan.py): clean.
Second session of round 3 (dense f64 decoder path, bitop3 complement, backward f32 gather): seeds 51-54 x 600, all equal.
Fourth session of round 3 (64-bit rotl1 / rotr1 as two v_alignbit_b32): seeds 61-64 x 300, all equal.
Round 5 (EXEC-masked f64 compaction + dense path, swizzled f32 3D store rows, post-B3 reordering): seeds 141-144 x 400, all equal.
Round 6 (the model itself changed: inactive DPP source lanes, shuffle width; kernels unchanged): seeds 671-674 x 1500, all equal."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypothesis import HealthCheck, given, settings, seed
from tests import test_wavesim_fuzz as f
from oracle import oracle
from tests.util import same_bits
from tests.wavesim import sim

count = [0]
@seed(int(sys.argv[1]))
@settings(max_examples=int(sys.argv[2]), deadline=None, suppress_health_check=list(HealthCheck), database=None)
@given(f._cases())
def run(case):
    dtype, shape, sd, (cus, bpc), schedule = case
    data = f._patterned(shape, dtype, sd)
    want = oracle.compress(data)
    got = sim.compress(data, cus=cus, blocks_per_cu=bpc, schedule=schedule)
    assert len(got) == len(want) and np.array_equal(got, want), (np.dtype(dtype).name, shape, sd)
    assert same_bits(sim.decompress(want, dtype, shape, schedule=schedule), data), (np.dtype(dtype).name, shape, sd)
    count[0] += 1
run()
print("ok", count[0])
