"""Time the OpenMP port on a saved grid -- the `cpu_baseline` leg of bench.py (TEST INFRASTRUCTURE ONLY).

Run as a subprocess so the OpenMP runtime starts with a pinned, spinning thread team (OMP_PROC_BIND / OMP_WAIT_POLICY
must be set before libgomp is loaded); prints one JSON line.  usage: python -m oracle.timing <grid.npy> <threads> [budget_s]"""
import json
import os
import sys
import time

import numpy as np


def main():
    path, threads = sys.argv[1], int(sys.argv[2])
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    from oracle import oracle

    grid = np.load(path, mmap_mode=None)
    wdt = np.uint32 if grid.dtype.itemsize == 4 else np.uint64
    # spread the pages of all three buffers over the NUMA nodes (first touch by the OpenMP team)
    grid = oracle.parallel_empty_like(grid, threads, copy=True)
    sbuf = oracle.parallel_empty_like(np.empty(oracle.compressed_length_bound(grid.dtype, grid.shape), dtype=wdt), threads, copy=False)
    obuf = oracle.parallel_empty_like(np.empty(grid.shape, dtype=grid.dtype), threads, copy=False)
    stream = oracle.compress(grid, threads, out=sbuf)  # untimed: touches buffers, starts the thread team
    oracle.decompress(stream, grid.dtype, grid.shape, threads, out=obuf)
    tc, td = [], []
    t_start = time.perf_counter()
    while len(tc) < 12 and time.perf_counter() - t_start < budget:
        t0 = time.perf_counter()
        stream = oracle.compress(grid, threads, out=sbuf)
        t1 = time.perf_counter()
        oracle.decompress(stream, grid.dtype, grid.shape, threads, out=obuf)
        t2 = time.perf_counter()
        tc.append(t1 - t0)
        td.append(t2 - t1)
    ok = bool(np.array_equal(obuf.view(wdt), grid.view(wdt)))
    nb = grid.nbytes
    print(json.dumps({"threads": threads, "reps": len(tc), "roundtrip_ok": ok, "stream_words": int(len(stream)),
                      "compress_GBps_best": nb / min(tc) / 1e9, "compress_GBps_median": nb / float(np.median(tc)) / 1e9,
                      "decompress_GBps_best": nb / min(td) / 1e9, "decompress_GBps_median": nb / float(np.median(td)) / 1e9}))


if __name__ == "__main__":
    main()
