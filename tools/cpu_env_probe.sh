#!/bin/bash
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from ndzip_amd.synth import synth_numpy
np.save('/dev/shm/probe.npy', synth_numpy((256,512,512), np.float32, 1, 0xff))
PY
run() { echo -n "$1 | threads $2: "; env $1 python -m oracle.timing /dev/shm/probe.npy $2 6 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('comp med %.1f best %.1f | decomp med %.1f best %.1f (reps %d)' % (d['compress_GBps_median'], d['compress_GBps_best'], d['decompress_GBps_median'], d['decompress_GBps_best'], d['reps']))"; }
for th in 32 64 128; do
run "X=1" $th
run "OMP_WAIT_POLICY=ACTIVE" $th
run "OMP_PROC_BIND=close OMP_PLACES=cores" $th
run "OMP_PROC_BIND=spread OMP_PLACES=cores OMP_WAIT_POLICY=ACTIVE" $th
run "OMP_PROC_BIND=true GOMP_SPINCOUNT=100000" $th
done
rm /dev/shm/probe.npy; lscpu | grep -E "NUMA|Socket|Thread|Core" | head -8
