#!/usr/bin/env python
"""Thread scaling of the OpenMP port (oracle) on the host CPUs of the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from ndzip_amd.synth import synth_numpy
print("omp max threads", oracle.max_threads(), "affinity", len(os.sched_getaffinity(0)), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
a = synth_numpy((256, 512, 512), np.float32, 1, 0xFF)
sbuf = np.zeros(oracle.compressed_length_bound(a.dtype, a.shape), np.uint32)
obuf = np.zeros_like(a)
for th in (1, 8, 16, 32, 64, 128, 256):
    s = oracle.compress(a, th, out=sbuf); oracle.decompress(s, a.dtype, a.shape, th, out=obuf)
    t0 = time.perf_counter(); s = oracle.compress(a, th, out=sbuf); t1 = time.perf_counter(); oracle.decompress(s, a.dtype, a.shape, th, out=obuf); t2 = time.perf_counter()
    print(f"threads {th:4d}: compress {a.nbytes/(t1-t0)/1e9:7.2f} GB/s  decompress {a.nbytes/(t2-t1)/1e9:7.2f} GB/s")
