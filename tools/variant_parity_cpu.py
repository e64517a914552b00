#!/usr/bin/env python
"""CPU check of A/B library variants (ndzip_amd/_variants/<name>.so from tools/build_variant.sh) BEFORE they cost GPU time: the
variant's gfx950 code objects are executed by the instruction-level interpreter (tests/gfx950_exec.py; the functional model plays
the host side) on mixed dense / sparse / zero data of four profiles and must reproduce the oracle's streams bit for bit, with no
wait-state or waitcnt finding.  Test tooling only.   usage: tools/variant_parity_cpu.py winpub wg3 plainloads ..."""
import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gfx950_exec as gx
from tests.wavesim import build as simbuild, sim
from tests.test_gfx950_exec import _mixed
from oracle import oracle
from tests.util import same_bits
import os
_orig = gx.Bridge.kernel_named
def _named(self, host_name):
    k = _orig(self, host_name)
    if k is None:  # lab builds: the kernels take one more uint32 (the experiment flags, ignored unless built for ablation)
        k = _orig(self, host_name + "j")
    return k
gx.Bridge.kernel_named = _named
for v in sys.argv[1:]:
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ndzip_amd', '_variants', f'{v}.so')
    b = gx.Bridge(simbuild.build(), [lib], tempfile.mkdtemp(prefix='gfxv'))
    for shape, dt in (((32, 32, 64), np.float32), ((16, 48, 32), np.float32), ((128, 192), np.float64), ((3*4096,), np.float32),
                      ((16, 32, 48), np.float64), ((2*4096 + 9,), np.float64), ((70, 130), np.float64)):  # (every 64-bit stencil: the f64sched variant rewrites them)
        data = _mixed(shape, dt, 7)
        want = oracle.compress(data)
        with b:
            got = sim.compress(data, cus=2, blocks_per_cu=2)
            back = sim.decompress(want, data.dtype, data.shape)
        assert np.array_equal(got, want), (v, shape)
        assert same_bits(back, data)
    print(v, 'ok', 'hazards', len(gx.HAZARD_LOG), 'waits', len(gx.WAIT_LOG))
