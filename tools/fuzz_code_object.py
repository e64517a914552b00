"""Extended property test of the BUILT gfx950 code objects on the instruction-level interpreter (tests/gfx950_exec.py; CPU only):
`tools/fuzz_code_object.py <seed> <cases>` runs `cases` random extents / value types / data mixtures (the generators of
tests/test_wavesim_fuzz.py) through the kernels as hipcc compiled them -- the host side on the functional model, every kernel launch
interpreted from libndzip_hip.so -- against the oracle: streams bit for bit, decompress(compress(x)) == x, with a random instruction
quantum for the interleaving of the workgroups in flight and either f64 decoder mapping.  Prints `ok <n>` and the hazard-log size."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypothesis import HealthCheck, given, seed, settings  # noqa: E402

from ndzip_amd import hip  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import gfx950_exec as gx  # noqa: E402
from tests import test_wavesim_fuzz as f  # noqa: E402
from tests.util import same_bits  # noqa: E402
from tests.wavesim import build as simbuild  # noqa: E402
from tests.wavesim import sim  # noqa: E402

bridge = gx.Bridge(simbuild.build(), [hip.LIB_PATH], tempfile.mkdtemp(prefix="gfx950_fuzz"))
real = gx.run_grid
count = [0]


@seed(int(sys.argv[1]))
@settings(max_examples=int(sys.argv[2]), deadline=None, suppress_health_check=list(HealthCheck), database=None)
@given(f._cases())
def run(case):
    dtype, shape, sd, (cus, bpc), _schedule = case
    if int(np.prod(shape)) > 40 * 4096:  # (keep a case at a few seconds)
        return
    rng = np.random.default_rng(sd)
    quantum = int(rng.choice([23, 150, 901, 4000]))
    cus = min(cus, 4)
    gx.run_grid = lambda *a, **k: real(*a, **{**k, "quantum": quantum})
    data = f._patterned(shape, dtype, sd)
    want = oracle.compress(data)
    with bridge:
        got = sim.compress(data, cus=cus, blocks_per_cu=2)
        back = sim.decompress(want, dtype, shape, f64_work_items=int(rng.choice([0, 128])))
    assert len(got) == len(want) and np.array_equal(got, want), (np.dtype(dtype).name, shape, sd, quantum)
    assert same_bits(back, data), (np.dtype(dtype).name, shape, sd)
    count[0] += 1


run()
print("ok", count[0], "hazards", len(gx.HAZARD_LOG))
