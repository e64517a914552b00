#!/bin/bash
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from ndzip_amd.synth import synth_numpy
np.save('/dev/shm/probe.npy', synth_numpy((512,512,512), np.float32, 1, 0xff))
PY
cat > /tmp/t.py <<'PY'
import sys, time, numpy as np, os
sys.path.insert(0, os.getcwd())
from oracle import oracle
g = np.load('/dev/shm/probe.npy'); th = int(sys.argv[1])
sbuf = np.zeros(oracle.compressed_length_bound(g.dtype, g.shape), np.uint32); obuf = np.zeros_like(g)
s = oracle.compress(g, th, out=sbuf); oracle.decompress(s, g.dtype, g.shape, th, out=obuf)
tc=[]; td=[]
for i in range(10):
    t0=time.perf_counter(); s = oracle.compress(g, th, out=sbuf); t1=time.perf_counter(); oracle.decompress(s, g.dtype, g.shape, th, out=obuf); t2=time.perf_counter()
    tc.append(g.nbytes/(t1-t0)/1e9); td.append(g.nbytes/(t2-t1)/1e9)
print('comp', ' '.join(f'{x:.0f}' for x in tc)); print('deco', ' '.join(f'{x:.0f}' for x in td))
PY
echo "== close/cores OMP_NUM_THREADS=128"; OMP_NUM_THREADS=128 OMP_PROC_BIND=close OMP_PLACES=cores python /tmp/t.py 128
echo "== close/cores, no OMP_NUM_THREADS"; OMP_PROC_BIND=close OMP_PLACES=cores python /tmp/t.py 128
echo "== close/threads 128"; OMP_PROC_BIND=close OMP_PLACES=threads python /tmp/t.py 128
echo "== no binding"; python /tmp/t.py 128
echo "== close/cores 64"; OMP_PROC_BIND=close OMP_PLACES=cores python /tmp/t.py 64
rm /dev/shm/probe.npy
