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

# persistent pipelined offloader, pinned buffers: H2D of job j+1, kernels of job j and D2H of job j-1 overlap
for slots in (1, 2, 3):
    jobs = 8
    p = ndzip_amd.HipPipelinedOffloader(np.float32, shape, slots=slots)
    bound = ndzip_amd.compressed_length_bound(np.float32, shape)
    ins = [ndzip_amd.PinnedBuffer(data.nbytes, np.float32) for _ in range(slots)]
    outs = [ndzip_amd.PinnedBuffer(bound * 4, np.uint32) for _ in range(slots)]
    for b in ins:
        b.array[:] = data.reshape(-1)
    lens = []
    for rep in range(2):
        t0 = time.perf_counter()
        for j in range(jobs):
            if j >= slots:
                lens.append(p.wait(j % slots)[0])
            p.submit_compress(j % slots, ins[j % slots].array.reshape(shape), outs[j % slots].array)
        for j in range(max(0, jobs - slots), jobs):
            lens.append(p.wait(j % slots)[0])
        t1 = time.perf_counter()
    words = lens[-1]
    streams = [o.array[:words] for o in outs]
    for rep in range(2):
        t2 = time.perf_counter()
        for j in range(jobs):
            if j >= slots:
                p.wait(j % slots)
            p.submit_decompress(j % slots, streams[j % slots], ins[j % slots].array.reshape(shape))
        for j in range(max(0, jobs - slots), jobs):
            p.wait(j % slots)
        t3 = time.perf_counter()
    ok = np.array_equal(ins[0].array.view(np.uint32), data.reshape(-1).view(np.uint32))
    print(f"pipelined offloader, {slots} slot(s), pinned, {jobs} x 512^3 f32: compress {jobs * data.nbytes / (t1 - t0) / 1e9:.2f} GB/s wall, "
          f"decompress {jobs * data.nbytes / (t3 - t2) / 1e9:.2f} GB/s wall; exact {ok}")
    p.close()
    for b in ins + outs:
        b.close()
