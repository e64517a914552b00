#!/usr/bin/env python
"""Where the s_nop of a compress kernel are EXECUTED (test tooling; tests/gfx950_exec.py interprets the built code object): wait states per
hypercube by the instruction in front of each s_nop -- dependent DPP steps, VCC hand-offs, asm statement boundaries, SGPR-spill lanes.
usage: tools/nop_sources.py <library .so> <shape>      e.g. tools/nop_sources.py ndzip_amd/_variants/f64sched.so 32,32,64   (float64 data)"""
import collections, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndzip_amd import hip, synth
from oracle import oracle
from tests import gfx950_exec as gx
from tests.wavesim import build as simbuild, sim
lib = sys.argv[1]; shape = tuple(int(x) for x in sys.argv[2].split(','))
orig = gx.Bridge.kernel_named
gx.Bridge.kernel_named = lambda self, n: orig(self, n) or orig(self, n + "j")
pcs = collections.Counter(); texts = {}; prevs = {}
real = gx.check_hazards
last = {}
def hook(w, ins):
    key = id(w)
    if ins.op == 's_nop' and 'compress' in w.kernel_name:
        pcs[ins.addr] += 1; texts[ins.addr] = ins.text; prevs[ins.addr] = last.get(key, '')
    last[key] = ins.text
    return real(w, ins)
gx.check_hazards = hook
data = synth.synth_numpy(shape, np.float64, seed=1, noise_mask=0xFF)
want = oracle.compress(data)
nhc = hip.num_hypercubes(shape)
b = gx.Bridge(simbuild.build(), [os.path.abspath(lib)], tempfile.mkdtemp(prefix="nop"))
with b:
    got = sim.compress(data, cus=4, blocks_per_cu=2)
assert np.array_equal(got, want)
tot = sum(pcs.values())
print('s_nop executed per hypercube', tot / nhc)
agg = collections.Counter()
for a, n in pcs.items():
    agg[(prevs[a].split()[0] if prevs[a] else '', texts[a])] += n
for k, n in agg.most_common(25):
    print(f'{n / nhc:7.1f}', k)
