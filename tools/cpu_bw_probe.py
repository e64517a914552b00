import sys, time, os, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle
for mb in (64, 256, 537, 1024):
    a = np.ones(mb << 18, np.float32)
    for th in (1, 16, 64, 128):
        b = oracle.parallel_empty_like(a, th)
        t0 = time.perf_counter()
        for _ in range(5): b = oracle.parallel_empty_like(a, th)   # includes a fresh allocation + first touch
        t1 = time.perf_counter()
        L = oracle.lib()
        t2 = time.perf_counter()
        for _ in range(5): L.ndzip_oracle_parallel_copy(b.ctypes.data, a.ctypes.data, a.nbytes, th)   # warm pages
        t3 = time.perf_counter()
        print(f"{mb} MB threads {th}: alloc+copy {5*a.nbytes/(t1-t0)/1e9:.1f} GB/s, warm copy {5*a.nbytes/(t3-t2)/1e9:.1f} GB/s")
