#!/usr/bin/env python
"""bench.py -- headline benchmark of the ndzip block encode/decode path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`, one rank per GPU over RCCL.  With RANK / WORLD_SIZE in the
environment (the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) this process IS a
rank; typed as is with N > 1 it starts the N ranks itself the same way and passes rank 0's line through (launch_ranks).
One "step" = one pass of the hot path over the synthetic grid, inputs already resident in HBM: compress, then decompress
(`--compress-only` / `--decompress-only`: that half).  Rank 0 prints ONE JSON line; its `ranks` object says how many ranks
the process group saw, over which backend, and on which device each one ran.

Workloads (`--config`, BASELINE.json `configs`):
  2 (default)  3D float32 512x512x512 per GPU -- the configuration the metric is quoted on.  At N GPUs the grid is
               (512 N) x 512 x 512 cut into N z-slabs: weak scaling, per-GPU work fixed.
  1            1D float32 16 Mi elements per GPU (the reference's CPU-runnable case; its `-e cpu -T 1` figure is the
               `cpu_reference_serial_cfg1` leg, reported with every run).
  3            2D float64 8192 x 8192 per GPU (64-bit transpose path).
  4            3D float32, z-slabs of 256 x 1024 x 1024 per GPU: at N = 8 exactly 2048 x 1024 x 1024 (8 GiB) sharded with
               the RCCL offset scan; fewer GPUs take fewer slabs (weak scaling).
  5            3D float64, z-slabs of 128 x 1024 x 1024 per GPU, decompress-only: at N = 8 exactly 1024^3 (8 GiB).
  16gib        3D float64 2048 x 1024 x 1024 = 16 GiB (2^31 elements, the largest grid the format's uint32 counts allow for
               this target) split over the N ranks: STRONG scaling, the ">= 6x at 8 GPUs over 1 GPU" line of north_star.
`--shape/--dtype` override the per-GPU slab.  The only exchange at N > 1 is the all-gather of one length per rank + the
header all-gather (ndzip_amd/sharded.py); bodies stay resident.

`value` = uncompressed bytes that went through compress plus uncompressed bytes that came out of decompress, over all
ranks, divided by the wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).

Extra objects:
  roofline      the dominant kernel of the step (compress_kernel_db<float,D> / compress_kernel_wide<u64,D>; decompress_kernel / decompress_kernel_wide<D>
                for decompress-only runs): algorithmic bytes (raw + stream) per launch over the HIP-event
                duration of the launch on the stream it runs on; peak 8 TB/s (HBM3E spec).  The other kernel rides along.
  cpu_baseline  the GENUINE reference CPU codec (oracle/_ref, compiled from /root/reference; serial path -- its OpenMP path
                needs Boost, absent here) on the host's physical cores, one independent block of the same grid per thread;
                median of several repetitions with best and spread; value null when best/median > 2 (a disturbed host).
                Falls back to the repo's OpenMP port (kind "port") where oracle/_ref is absent.  The port, the serial
                reference on a sample of the workload and on config 1's 1D 16 Mi array ride along as extra legs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0        # measured float4 copy ceiling from the same guide

CONFIGS = {
    # name: (description, dtype, per-GPU slab or None, global extent for strong scaling or None, mode)
    "1": ("BASELINE configs[0]: 1D float32 16 Mi", "float32", (1 << 24,), None, "both"),
    "2": ("BASELINE configs[1]: 3D float32 512x512x512", "float32", (512, 512, 512), None, "both"),
    "3": ("BASELINE configs[2]: 2D float64 8192x8192", "float64", (8192, 8192), None, "both"),
    "4": ("BASELINE configs[3]: 3D float32 2048x1024x1024 over 8 GPUs = z-slabs of 256x1024x1024", "float32", (256, 1024, 1024), None, "both"),
    "5": ("BASELINE configs[4]: 3D float64 1024^3 over 8 GPUs = z-slabs of 128x1024x1024, decompress-only", "float64", (128, 1024, 1024), None,
          "decompress"),
    "16gib": ("north_star scaling target: 3D float64 2048x1024x1024 = 16 GiB, strong scaling", "float64", None, (2048, 1024, 1024), "both"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=str, default="2", choices=sorted(CONFIGS), help="BASELINE.json workload (see the module docstring)")
    ap.add_argument("--shape", type=str, default=None, help="override: per-GPU slab (dimension 0 is multiplied by --gpus)")
    ap.add_argument("--dtype", type=str, default=None, choices=["float32", "float64"])
    ap.add_argument("--noise-mask", type=lambda s: int(s, 0), default=0xFF)
    ap.add_argument("--smooth", action="store_true", help="highly compressible bookend (two low-frequency octaves)")
    ap.add_argument("--data", type=str, default="synthetic", choices=["synthetic", "random", "zeros"])
    ap.add_argument("--workgroups-per-cu", type=int, default=0, help="cap the persistent compress grid (0 = default: 4 per CU); 3 = the round-2 grid, for an A/B of the occupancy")
    ap.add_argument("--f64-work-items", type=int, default=0, choices=[0, 128, 256], help="64-bit decoder mapping: 0 = the library's default (128 work-items per hypercube, decompress_kernel, until the other is measured), 256 = decompress_kernel_wide; for an A/B of the two")
    ap.add_argument("--overlap-exchange", action="store_true", help="N > 1: leave the offset / header exchange in flight behind the decompress launch (opt-in until it has run over RCCL on a multi-GPU node; default: compress -> exchange -> decompress, every collective waited for)")
    ap.add_argument("--native-exchange", action="store_true", help="drive the step through the C++ host of the sharded path (libndzip_hip_rccl.so, include/ndzip_hip_sharded.h: plan, buffers and "
                    "the two all-gathers in C++ over its own ncclComm_t) instead of ndzip_amd.sharded.ShardedCodec (torch.distributed); same kernels, same stream -- opt-in until it has run on a multi-GPU node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work per cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--compress-only", action="store_true", help="a step is the compress half only")
    ap.add_argument("--decompress-only", action="store_true", help="a step is the decompress half only (the grid is compressed once, untimed)")
    ap.add_argument("--lib", type=str, default=None, help="A/B tooling: load this build of the library instead of ndzip_amd/libndzip_hip.so")
    return ap.parse_args(argv)


# ---- CPU legs (rank 0, N = 1 only; bounded samples; the checker's code is timed here, never shipped) -------------------------

def physical_cores() -> int:
    import subprocess

    cores = os.cpu_count() or 1
    try:  # physical cores = the reference's default thread count (cpu_factory.cc:8-9)
        txt = subprocess.run(["lscpu", "-p=CORE,SOCKET"], capture_output=True, text=True).stdout
        phys = len({l for l in txt.splitlines() if l and not l.startswith("#")})
        if phys > 0:
            cores = min(cores, phys)
    except Exception:
        pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return max(1, cores)


def _blocks(shape, threads):
    """Cut `shape` into up to `threads` blocks of whole hypercubes: dimension 0 first, then dimension 1.  Every block is an
    independent ndzip array.  -> list of tuples of slices"""
    side = {1: 4096, 2: 64, 3: 16}[len(shape)]
    g0 = shape[0] // side
    n0 = max(1, min(threads, g0))
    n1 = 1
    if len(shape) >= 2:
        g1 = shape[1] // side
        n1 = max(1, min(threads // n0, g1))
    out = []
    for i in range(n0):
        a, b = i * g0 // n0 * side, (i + 1) * g0 // n0 * side
        if len(shape) == 1:
            out.append((slice(a, b),))
            continue
        for j in range(n1):
            c, d = j * g1 // n1 * side, (j + 1) * g1 // n1 * side
            out.append((slice(a, b), slice(c, d)))
    return [s for s in out if all(x.stop > x.start for x in s)]


def _spread_stats(times, nbytes):
    import numpy as np

    med, best = float(np.median(times)), float(min(times))
    return nbytes / med / 1e9, nbytes / best / 1e9, med / best


def cpu_reference_blocks(host_grid, cores, budget_s):
    """The genuine reference serial codec (oracle/_ref), one independent block of the grid per thread, all threads at once:
    what the host's cores give a caller of the reference without its Boost-dependent OpenMP back-end.  ctypes releases the
    GIL around the calls; every buffer is first touched by the thread that uses it.  Repeated until the budget is spent
    (at least 3, at most 15 repetitions); per repetition the wall time of the slowest thread counts."""
    import threading

    import numpy as np

    from oracle import oracle

    blocks = _blocks(host_grid.shape, cores)
    threads = len(blocks)
    wdt = np.uint32 if host_grid.itemsize == 4 else np.uint64
    max_reps = 15
    tc = np.zeros((threads, max_reps))
    td = np.zeros((threads, max_reps))
    ok = [False] * threads
    stop = [False]
    bar = threading.Barrier(threads + 1, timeout=180)  # (timeout + abort: a failing worker must not hang the benchmark)

    def work(i):
        try:
            local = np.array(host_grid[blocks[i]], copy=True)   # first touch on this thread
            s = oracle.ref_compress(local)                       # warm code and buffers
            back, _ = oracle.ref_decompress(s, local.dtype, local.shape)
            ok[i] = bool(np.array_equal(back.view(wdt).reshape(-1), local.view(wdt).reshape(-1)))
            r = 0
            while True:
                bar.wait()
                if stop[0]:
                    break
                t0 = time.perf_counter()
                s = oracle.ref_compress(local)
                t1 = time.perf_counter()
                bar.wait()
                t2 = time.perf_counter()
                oracle.ref_decompress(s, local.dtype, local.shape)
                t3 = time.perf_counter()
                tc[i, r], td[i, r] = t1 - t0, t3 - t2
                r += 1
                bar.wait()
        except Exception:
            bar.abort()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    wall_c, wall_d = [], []
    t_begin = time.perf_counter()
    while len(wall_c) < max_reps and (len(wall_c) < 3 or time.perf_counter() - t_begin < budget_s):
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        t1 = time.perf_counter()
        bar.wait()
        t2 = time.perf_counter()
        wall_c.append(t1 - t0)
        wall_d.append(t2 - t1)
    stop[0] = True
    bar.wait()
    for t in ts:
        t.join()
    nbytes = sum(host_grid[b].size for b in blocks) * host_grid.itemsize
    c_med, c_best, c_spread = _spread_stats(wall_c, nbytes)
    d_med, d_best, d_spread = _spread_stats(wall_d, nbytes)
    stable = c_spread <= 2.0 and d_spread <= 2.0
    leg = {
        "value": round(2.0 / (1.0 / c_med + 1.0 / d_med), 3) if stable else None,
        "unit": "GB/s",
        "cores": threads,
        "kind": "reference",
        "sample": f"{threads} blocks of the {'x'.join(map(str, host_grid.shape))} {host_grid.dtype} grid ({nbytes >> 20} MiB in all), each compressed "
                  f"and decompressed as its own array by the reference's serial CPU codec (oracle/_ref, compiled from the reference "
                  f"sources) on its own thread, all threads at once; median of {len(wall_c)} repetitions, wall time of the slowest thread",
        "compress_GBps": round(c_med, 3),
        "decompress_GBps": round(d_med, 3),
        "compress_GBps_best": round(c_best, 3),
        "decompress_GBps_best": round(d_best, 3),
        "median_over_best": [round(c_spread, 2), round(d_spread, 2)],
        "roundtrip_ok": all(ok),
    }
    if not stable:
        leg["reason"] = "best and median repetition differ by more than 2x: the host was disturbed, no value reported"
    return leg


def cpu_port_openmp(host_grid, cores, budget_s):
    """The repo's OpenMP restatement of the reference CPU codec (oracle/ndzip_oracle.c, bit-exact with the compiled reference),
    timed in a subprocess (oracle/timing.py) so its thread team is pinned and spinning."""
    import subprocess
    import tempfile

    import numpy as np

    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, f"ndzip_bench_grid_{os.getpid()}.npy")
    np.save(path, host_grid)
    try:
        # close binding on physical cores was the stable setting on the 2 x 64-core host (round-1 measurement, profiles/README.md)
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
        env.pop("OMP_WAIT_POLICY", None)
        r = subprocess.run([sys.executable, "-m", "oracle.timing", path, str(cores), str(budget_s)], capture_output=True, text=True, cwd=ROOT,
                           env=env, timeout=300)
        t = json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        os.remove(path)
    c, d = t["compress_GBps_median"], t["decompress_GBps_median"]
    spread = [t["compress_GBps_best"] / c, t["decompress_GBps_best"] / d]
    stable = max(spread) <= 2.0
    leg = {
        "value": round(2.0 / (1.0 / c + 1.0 / d), 3) if stable else None,
        "unit": "GB/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{'x'.join(map(str, host_grid.shape))} {host_grid.dtype} grid, median of {t['reps']} reps compress+decompress, OpenMP port of the "
                  f"reference CPU codec (oracle/ndzip_oracle.c: scalar C, the reference is AVX2), threads pinned to physical cores",
        "compress_GBps": round(c, 3),
        "decompress_GBps": round(d, 3),
        "compress_GBps_best": round(t["compress_GBps_best"], 3),
        "decompress_GBps_best": round(t["decompress_GBps_best"], 3),
        "median_over_best": [round(x, 2) for x in spread],
        "roundtrip_ok": t["roundtrip_ok"],
    }
    if not stable:
        leg["reason"] = "best and median repetition differ by more than 2x: the host was disturbed, no value reported"
    return leg


def cpu_reference_serial(sample, what):
    """The genuine reference serial path, one thread (the reference tool's `-e cpu -T 1`, src/compress/compress.cc:103-107)."""
    import numpy as np

    from oracle import oracle

    oracle.ref_decompress(oracle.ref_compress(sample), sample.dtype, sample.shape)  # untimed: code and pages warm
    tc, td = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        s = oracle.ref_compress(sample)
        t1 = time.perf_counter()
        back, _ = oracle.ref_decompress(s, sample.dtype, sample.shape)
        t2 = time.perf_counter()
        tc.append(t1 - t0)
        td.append(t2 - t1)
    wdt = np.uint32 if sample.itemsize == 4 else np.uint64
    c, d = sample.nbytes / float(np.median(tc)) / 1e9, sample.nbytes / float(np.median(td)) / 1e9
    return {
        "value": round(2.0 / (1.0 / c + 1.0 / d), 3),
        "unit": "GB/s",
        "cores": 1,
        "kind": "reference",
        "sample": f"{what} ({sample.nbytes >> 20} MiB), reference serial CPU path compiled from the reference sources (oracle/_ref), median of 3",
        "compress_GBps": round(c, 3),
        "decompress_GBps": round(d, 3),
        "ratio": round(len(s) * s.itemsize / sample.nbytes, 4),
        "roundtrip_ok": bool(np.array_equal(back.view(wdt).reshape(-1), sample.view(wdt).reshape(-1))),
    }


def cpu_port_serial(sample, what):
    """The C restatement of the reference CPU codec (oracle/ndzip_oracle.c), one thread: what stands in for the reference tool's
    `-e cpu -T 1` when oracle/_ref did not reach the box."""
    import numpy as np

    from oracle import oracle

    oracle.decompress(oracle.compress(sample, 1), sample.dtype, sample.shape, 1)  # untimed: code and pages warm
    tc, td = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        s = oracle.compress(sample, 1)
        t1 = time.perf_counter()
        back = oracle.decompress(s, sample.dtype, sample.shape, 1)
        back = back[0] if isinstance(back, tuple) else back
        t2 = time.perf_counter()
        tc.append(t1 - t0)
        td.append(t2 - t1)
    wdt = np.uint32 if sample.itemsize == 4 else np.uint64
    c, d = sample.nbytes / float(np.median(tc)) / 1e9, sample.nbytes / float(np.median(td)) / 1e9
    return {
        "value": round(2.0 / (1.0 / c + 1.0 / d), 3),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{what} ({sample.nbytes >> 20} MiB), C restatement of the reference serial CPU path (oracle/ndzip_oracle.c: scalar C, the "
                  f"reference is AVX2), one thread, median of 3",
        "compress_GBps": round(c, 3),
        "decompress_GBps": round(d, 3),
        "ratio": round(len(s) * s.itemsize / sample.nbytes, 4),
        "roundtrip_ok": bool(np.array_equal(np.asarray(back).view(wdt).reshape(-1), sample.view(wdt).reshape(-1))),
    }


def cpu_legs(host_grid, budget_s):
    """-> dict of the CPU objects of the JSON line.  `host_grid`: a bounded sample of the benchmarked workload."""
    import numpy as np

    from ndzip_amd.synth import synth_numpy
    from oracle import oracle

    out = {}
    cores = physical_cores()

    def guarded(name, fn, kind):
        try:
            out[name] = fn()
        except Exception as e:  # a reported extra must never cost the GPU number
            out[name] = {"value": None, "unit": "GB/s", "cores": 0, "kind": kind, "sample": f"failed: {type(e).__name__}: {e}"}

    # which checker is timed, and why: the compiled reference (oracle/_ref/libndzip_ref.so, built from /root/reference where that
    # exists and shipped as a binary) when it is there, else the C restatement (oracle/libndzip_oracle.so)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libndzip_ref.so")
    why = (f"kind=reference: {os.path.relpath(ref_so, ROOT)} is present" if oracle.have_ref()
           else f"kind=port: {os.path.relpath(ref_so, ROOT)} is absent (not built here and not shipped), timing the C restatement")
    print(f"[bench] cpu_baseline {why}", file=sys.stderr, flush=True)
    if oracle.have_ref():
        guarded("cpu_baseline", lambda: cpu_reference_blocks(host_grid, cores, budget_s), "reference")
        guarded("cpu_port_openmp", lambda: cpu_port_openmp(host_grid, cores, min(budget_s, 8.0)), "port")
        first = host_grid[: max(16, host_grid.shape[0] // 8)] if host_grid.ndim == 3 else host_grid[: max(1, host_grid.shape[0] // 8)]
        guarded("cpu_reference_serial", lambda: cpu_reference_serial(np.ascontiguousarray(first), f"first {first.shape[0]} rows of dimension 0 of the workload"),
                "reference")
        cfg1 = synth_numpy((1 << 24,), np.float32, seed=3, noise_mask=0xFF)
        guarded("cpu_reference_serial_cfg1",
                lambda: cpu_reference_serial(cfg1, "BASELINE configs[0]: 1D float32 16 Mi elements (Appendix-B generator seed 3, noise_mask 0xff)"),
                "reference")
    else:
        guarded("cpu_baseline", lambda: cpu_port_openmp(host_grid, cores, budget_s), "port")
        # BASELINE configs[0] has a line either way: with the port when the compiled reference is not there
        cfg1 = synth_numpy((1 << 24,), np.float32, seed=3, noise_mask=0xFF)
        guarded("cpu_port_serial_cfg1",
                lambda: cpu_port_serial(cfg1, "BASELINE configs[0]: 1D float32 16 Mi elements (Appendix-B generator seed 3, noise_mask 0xff)"), "port")
    if isinstance(out.get("cpu_baseline"), dict):
        out["cpu_baseline"]["why_kind"] = why
    return out


# ---- the GPU benchmark ------------------------------------------------------------------------------------------------------

class Accelerator:
    """The four things main() asks of torch.cuda, in one place (tests/test_bench_cpu.py substitutes a host stand-in to run
    main() end to end against the kernels' functional model; the benchmark itself always runs on the GPU)."""

    def __init__(self, index):
        import torch

        assert torch.cuda.is_available(), "bench.py needs a GPU: the ndzip HIP back-end has no CPU fallback"
        torch.cuda.set_device(index)
        self.device = torch.device("cuda", index)

    def synchronize(self):
        import torch

        torch.cuda.synchronize()

    def event(self):
        import torch

        return torch.cuda.Event(enable_timing=True)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` typed as is (no RANK / WORLD_SIZE in the environment): start the N ranks ourselves, exactly the
    way the driver would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port
    <free> bench.py <same arguments>` -- pass rank 0's JSON line through to our stdout (everything else the ranks or the launcher
    print goes to stderr, so stdout carries the ONE line of the contract) and return the launcher's exit status.
    NDZIP_BENCH_ENTRY names the script the ranks run instead of this file (tests/bench_on_model.py: the same main() with the
    kernels on the functional model, which is how the CPU suite exercises this launch without a GPU)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    entry = os.environ.get("NDZIP_BENCH_ENTRY") or os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only kind the host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, physical_cores() // args.gpus)))
    env["NDZIP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), entry] + list(sys.argv[1:] if argv is None else argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    lines = 0
    for line in proc.stdout:
        is_result = line.startswith('{"metric"')
        lines += is_result
        (sys.stdout if is_result else sys.stderr).write(line)
        (sys.stdout if is_result else sys.stderr).flush()
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks exited 0 but printed {lines} result lines", file=sys.stderr)
        rc = 1
    return rc


def main(argv=None):
    args = parse_args(argv)
    if args.compress_only and args.decompress_only:
        raise SystemExit("--compress-only and --decompress-only exclude each other")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = launch_ranks(args, argv)
        if rc:
            raise SystemExit(rc)
        return
    import numpy as np
    import torch
    import torch.distributed as dist

    from ndzip_amd import hip

    if args.lib:  # (a variant may be a build of an earlier commit that lacks the newest entry points: bind what it has)
        import ctypes

        import torch  # noqa: F401  (first, so the library binds to the HIP runtime torch loaded)

        hip._lib = hip._bind(ctypes.CDLL(os.path.abspath(args.lib)), strict=False)
    import ndzip_amd
    from ndzip_amd.sharded import ShardedCodec, plan_shards
    from ndzip_amd.synth import synth_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NDZIP_BENCH_SHARE_GPU=1 (self-test on a one-GPU box only): every rank uses cuda:0 and the group is gloo -- RCCL
    # refuses two ranks on one device; same code path otherwise
    share_gpu = world > 1 and os.environ.get("NDZIP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share_gpu else local_rank
    acc = Accelerator(dev_index)
    device = acc.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    desc, cfg_dtype, cfg_slab, cfg_global, cfg_mode = CONFIGS[args.config]
    np_dtype = np.dtype(args.dtype or cfg_dtype)
    if args.shape:
        per_gpu = tuple(int(x) for x in args.shape.split(","))
        global_extent = (per_gpu[0] * world,) + per_gpu[1:]
        scaling, desc = "weak", f"custom slab {args.shape}"
    elif cfg_global is not None:
        global_extent = cfg_global
        scaling = "strong"
    else:
        per_gpu = cfg_slab
        global_extent = (per_gpu[0] * world,) + per_gpu[1:]
        scaling = "weak"
    mode = "compress" if args.compress_only else "decompress" if (args.decompress_only or cfg_mode == "decompress") else "both"
    dims = len(global_extent)
    t_dtype = torch.float32 if np_dtype == np.float32 else torch.float64
    # N > 1: the offset / header exchange (all-gather of one length per rank, base + global offsets, all-gather of the header
    # segments: 8 x 128 KiB at N = 8) lies between the two launches: compress -> exchange -> decompress, every collective waited
    # for.  --overlap-exchange lets it run BEHIND the decompress launch instead, which decodes the rank's slab from its local
    # offsets and needs nothing from the other ranks (ndzip_amd/sharded.py: overlap_exchange) -- rehearsed over gloo on the
    # functional model and in tests/test_hip_sharded_rccl.py, opt-in until that test has passed on a multi-GPU node.
    if args.native_exchange:
        if args.workgroups_per_cu or args.f64_work_items:
            raise SystemExit("--native-exchange has no --workgroups-per-cu / --f64-work-items (A/B handles of the Python driver)")
        from ndzip_amd import sharded_native

        codec = sharded_native.NativeShardedCodec(np_dtype, global_extent, rank, world, device, overlap_exchange=world > 1 and args.overlap_exchange)
    else:
        codec = ShardedCodec(np_dtype, global_extent, rank, world, device, overlap_exchange=world > 1 and args.overlap_exchange)
    if args.workgroups_per_cu:
        codec.compressor.set_max_workgroups_per_cu(args.workgroups_per_cu)
    if args.f64_work_items:
        codec.decompressor.set_f64_work_items(args.f64_work_items)
    shard = codec.shard

    # ---- synthetic input, generated directly in HBM (identical bits on every machine) ------------------------------
    if args.data == "synthetic":
        # the slab's values depend on the GLOBAL coordinates: generate with the global linear offset
        local = synth_slab(synth_torch, global_extent, shard, t_dtype, device, args.noise_mask, args.smooth)
    elif args.data == "random":
        g = torch.Generator(device=device)
        g.manual_seed(1234 + rank)
        it = torch.int32 if np_dtype == np.float32 else torch.int64
        local = torch.randint(-2 ** 31, 2 ** 31 - 1, shard.extent, dtype=torch.int32, device=device, generator=g).view(torch.float32) \
            if np_dtype == np.float32 else torch.randint(-2 ** 62, 2 ** 62, shard.extent, dtype=it, device=device, generator=g).view(torch.float64)
    else:
        local = torch.zeros(shard.extent, dtype=t_dtype, device=device)
    out = torch.empty_like(local)
    raw_bytes_local = local.numel() * local.element_size()

    if mode == "decompress":  # the stream that every timed step decodes
        codec.compress(local)
        acc.synchronize()
        codec.check()

    def step(ev=None):
        # ev[0..1] bracket the compress launch (compress kernel [+ border kernel]) on the stream it runs on, ev[2..3] the
        # decompress launch (recorded inside decompress(), right around the decode launch: with --overlap-exchange the rest of
        # the exchange is enqueued behind it and is not part of the bracket); the offset / header exchange of the N > 1 path
        # lies between ev[1] and ev[2] by default
        if mode != "decompress":
            codec.compress(local, kernel_events=(ev[0], ev[1]) if ev else None)
        if mode != "compress":
            codec.decompress(out, kernel_events=(ev[2], ev[3]) if ev else None)

    for _ in range(args.warmup):
        step()
    acc.synchronize()
    codec.check()

    events = [[acc.event() for _ in range(4)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    acc.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    acc.synchronize()
    if world > 1:
        dist.barrier()
    acc.synchronize()
    elapsed = time.perf_counter() - t0
    codec.check()

    t_comp = None if mode == "decompress" else sum(e[0].elapsed_time(e[1]) for e in events) / args.steps * 1e-3   # seconds per launch
    t_decomp = None if mode == "compress" else sum(e[2].elapsed_time(e[3]) for e in events) / args.steps * 1e-3

    # ---- verification (outside the timed region): round trip is bit-exact ------------------------------------------
    body_len = codec.body_words() if args.native_exchange else int(codec.body_len.cpu()[0]) & 0xFFFFFFFF
    ok = True
    if not args.no_verify and mode != "compress":
        it = torch.int32 if np_dtype == np.float32 else torch.int64
        ok = bool(torch.equal(out.view(it), local.view(it)))
    stats = torch.tensor([elapsed, t_comp or 0.0, t_decomp or 0.0, float(body_len), float(ok)], dtype=torch.float64, device=device)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        mn = stats.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        elapsed = float(mx[0])
        t_comp = float(mx[1]) if t_comp is not None else None
        t_decomp = float(mx[2]) if t_decomp is not None else None
        total_body_words = float(sm[3])
        ok = bool(mn[4] > 0.5)
        ids = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([dev_index], dtype=torch.int64, device=device))
        ranks_seen = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "device_ids": [int(t[0]) for t in ids],
                      "launched_by": "bench.py (torch.distributed.run child)" if os.environ.get("NDZIP_BENCH_SELF_LAUNCHED") else "torch.distributed.run"}
    else:
        total_body_words = float(body_len)
        ranks_seen = {"world_size": 1, "backend": None, "device_ids": [dev_index], "launched_by": "python"}

    if rank == 0:
        wb = np_dtype.itemsize
        nhc_total = ndzip_amd.num_hypercubes(global_extent)
        raw_total = float(np.prod(global_extent, dtype=np.float64)) * wb
        stream_bytes_total = (ndzip_amd.header_words(np_dtype, nhc_total) + total_body_words) * wb
        ratio = stream_bytes_total / raw_total
        passes = 2 if mode == "both" else 1
        value = passes * raw_total * args.steps / elapsed / 1e9
        # per GPU and launch: N read + C written (compress), C read + N written (decompress) -- SURVEY 8d
        algo_bytes_per_launch = (raw_total + stream_bytes_total) / world
        slabs = plan_shards(global_extent, world)
        kernel_c = f"compress_kernel_db<float,{dims}>" if wb == 4 else f"compress_kernel_wide<unsigned long,{dims}>"
        kernel_d = (f"decompress_kernel<{'float' if wb == 4 else 'double'},{dims}>" if wb == 4 or args.f64_work_items != 256
                    else f"decompress_kernel_wide<{dims}>")

        def leg(t):
            a = algo_bytes_per_launch / t / 1e9
            return {"achieved": round(a, 2), "frac": round(a / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(a / HBM_COPY_GBPS, 4),
                    "launch_ms": round(t * 1e3, 4), "uncompressed_GBps": round(raw_total / world / t / 1e9, 2)}

        dominant_t, dominant_k = (t_decomp, kernel_d) if mode == "decompress" else (t_comp, kernel_c)
        roofline = {"bound": "hbm", "kernel": dominant_k, **leg(dominant_t), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
                    "algorithmic_bytes_per_launch": int(algo_bytes_per_launch)}
        if mode == "both":
            roofline["decompress"] = {"kernel": kernel_d, **leg(t_decomp)}
        result = {
            "metric": "compress + decompress GB/s (uncompressed) per GPU; % of HBM3E peak",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32" if wb == 4 else "u64",
            "data": "synthetic" if args.data == "synthetic" else args.data,
            "config": {
                "workload": f"{desc}: {dims}D {np_dtype.name} {'x'.join(map(str, global_extent))} synthetic turbulence "
                            f"(Appendix-B generator seed 1, noise_mask {args.noise_mask:#x}{', smooth' if args.smooth else ''}); "
                            f"{world} z-slab(s) of {'x'.join(map(str, slabs[0].extent))}",
                "baseline_config": args.config if not args.shape else None,
                "hypercubes": nhc_total,
                "compression_ratio": round(ratio, 4),
                "step": {"both": "compress then decompress", "compress": "compress only", "decompress": "decompress only"}[mode]
                        + ", inputs resident in HBM",
                "parallelism": f"hypercube-range sharding x{world}" + ((" (RCCL all-gather of offsets + header" + (", behind the decompress launch)" if args.overlap_exchange else ")")) if world > 1 else ""),
                "host": "C++ (libndzip_hip_rccl.so: ndzip_hip_sharded_*, RCCL called from C++)" if args.native_exchange else "Python (ndzip_amd.sharded.ShardedCodec over torch.distributed)",
            },
            "per_gpu": {},
            "ranks": ranks_seen,
            "roofline": roofline,
            "roundtrip_bit_exact": ok,
        }
        if args.lib:
            result["config"]["lib"] = args.lib
        if t_comp is not None:
            result["per_gpu"]["compress_GBps"] = round(raw_total / world / t_comp / 1e9, 2)
            result["per_gpu"]["compress_frac_of_hbm_peak"] = round(raw_total / world / t_comp / 1e9 / HBM_PEAK_GBPS, 4)
        if t_decomp is not None:
            result["per_gpu"]["decompress_GBps"] = round(raw_total / world / t_decomp / 1e9, 2)
            result["per_gpu"]["decompress_frac_of_hbm_peak"] = round(raw_total / world / t_decomp / 1e9 / HBM_PEAK_GBPS, 4)
        # HBM bytes per launch from the PMC passes (tools/pmc.sh -> profiles/traffic.json): only counters that were taken on
        # THESE kernels (fingerprint of the device-code sources) may stand next to this run's launch time; anything else is null
        result["roofline"]["traffic"] = None
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file) and not args.lib:
            try:
                from ndzip_amd.build import kernels_fingerprint

                with open(traffic_file) as f:
                    tr = json.load(f)
                key = f"{np_dtype.name}-{'x'.join(map(str, slabs[0].extent))}"
                if key in tr:
                    have, built = tr[key].get("kernels"), kernels_fingerprint()
                    if have == built:
                        which = "decompress_hbm_bytes_per_launch" if mode == "decompress" else "compress_hbm_bytes_per_launch"
                        result["roofline"]["traffic"] = tr[key].get(which)
                        result["roofline"]["traffic_source"] = tr[key].get("source")
                    else:
                        result["roofline"]["traffic_source"] = (f"none: profiles/traffic.json holds counters of kernels {have}, "
                                                                f"this tree's are {built} (re-run tools/pmc.sh)")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            # a bounded sample of the same workload: at most 512 MiB of leading rows of dimension 0 (whole hypercube planes)
            side = {1: 4096, 2: 64, 3: 16}[dims]
            row_bytes = raw_bytes_local // max(1, local.shape[0])
            rows = min(local.shape[0], max(side, (512 << 20) // max(1, row_bytes) // side * side))
            try:
                result.update(cpu_legs(local[:rows].cpu().numpy(), args.cpu_budget))
            except Exception as e:  # the checker is optional for the GPU number, never the other way round
                result["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


def synth_slab(synth_torch, global_extent, shard, t_dtype, device, noise_mask, smooth):
    """Generate this rank's z-slab of the global field (values depend on global coordinates and linear index)."""
    import torch

    from ndzip_amd.synth import synth_torch_range

    rest = 1
    for x in global_extent[1:]:
        rest *= x
    out = torch.empty(shard.extent, dtype=t_dtype, device=device)
    synth_torch_range(global_extent, t_dtype, shard.start0 * rest, shard.extent[0] * rest, out.view(-1), seed=1, noise_mask=noise_mask,
                      smooth=smooth)
    return out


if __name__ == "__main__":
    main()
