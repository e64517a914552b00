#!/usr/bin/env python
"""PCIe-inclusive throughput of the host-pointer offloader (H2D + kernels + D2H), for DESIGN.md; never bench.py's value."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ndzip_amd
from ndzip_amd.synth import synth_torch
import torch

shape = (512, 512, 512)
data = synth_torch(shape, torch.float32, 1, 0xFF, device="cuda").cpu().numpy()
off = ndzip_amd.make_hip_offloader(np.float32, 3)
off.compress(data[:64])
for _ in range(2):
    t0 = time.perf_counter(); s = off.compress(data); t1 = time.perf_counter()
    back, n = off.decompress(s, shape); t2 = time.perf_counter()
print(f"offloader 512^3 f32 (pageable host memory): compress {data.nbytes/(t1-t0)/1e9:.2f} GB/s wall (kernel {off.last_kernel_ns/1e6:.3f} ms), "
      f"decompress {data.nbytes/(t2-t1)/1e9:.2f} GB/s wall; exact {np.array_equal(back.view(np.uint32), data.view(np.uint32))}")
