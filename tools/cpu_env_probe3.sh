#!/bin/bash
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from ndzip_amd.synth import synth_numpy
np.save('/dev/shm/probe.npy', synth_numpy((512,512,512), np.float32, 1, 0xff))
PY
for th in 64 128; do for env in "X=1" "OMP_PROC_BIND=close OMP_PLACES=cores" "OMP_PROC_BIND=spread OMP_PLACES=cores"; do echo -n "$env threads $th: "; env $env OMP_NUM_THREADS=$th python -m oracle.timing /dev/shm/probe.npy $th 8 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('comp med %.1f best %.1f | decomp med %.1f best %.1f (reps %d)' % (d['compress_GBps_median'], d['compress_GBps_best'], d['decompress_GBps_median'], d['decompress_GBps_best'], d['reps']))"; done; done
rm /dev/shm/probe.npy
