#!/usr/bin/env python
"""bench.py -- headline benchmark of the ndzip block encode/decode path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by torch.distributed.run, one
rank per GPU over RCCL).  One "step" = compress the synthetic grid, then decompress it again, with the input
already resident in HBM.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 3D float32 512x512x512 synthetic turbulence per GPU (SURVEY.md Appendix B
generator, seed 1, noise_mask 0xff).  At N GPUs the grid is (512*N) x 512 x 512 cut into N z-slabs (weak scaling:
per-GPU work fixed); the only exchange is the RCCL all-gather of one length per rank + the header all-gather
(ndzip_amd/sharded.py).  `value` = uncompressed bytes that went through compress plus uncompressed bytes that
came out of decompress, over all ranks, divided by the wall time of the K timed steps (max over ranks).

Extra objects:
  roofline      dominant kernel = compress_kernel_db<float,3> (compress_kernel_wide<u64,D> for float64 runs): algorithmic bytes (raw in + stream out) per launch over
                the HIP-event duration of the launch on the stream it runs on; peak 8 TB/s (HBM3E spec).
  cpu_baseline  this repo's OpenMP port of the reference CPU path (oracle/, bit-exact with the compiled
                reference) timed on the host cores of the same box on the same grid; plus the genuine reference
                serial path (oracle/_ref) on a z-slab sample as `cpu_reference_serial`.
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", type=str, default="512,512,512", help="per-GPU slab (dimension 0 is multiplied by --gpus)")
    ap.add_argument("--dtype", type=str, default="float32", choices=["float32", "float64"])
    ap.add_argument("--noise-mask", type=lambda s: int(s, 0), default=0xFF)
    ap.add_argument("--smooth", action="store_true", help="highly compressible bookend (two low-frequency octaves)")
    ap.add_argument("--data", type=str, default="synthetic", choices=["synthetic", "random", "zeros"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--compress-only", action="store_true", help="timing experiments: skip the decompress half of a step")
    return ap.parse_args()


def cpu_baseline(host_grid, dims):
    """OpenMP port on the host cores + the genuine serial reference on a bounded sample (rank 0, N = 1 only).
    The port is timed in a subprocess (oracle/timing.py) so its OpenMP team is pinned and spinning; medians over up to
    12 repetitions of the full grid (about 20 s at the most)."""
    import subprocess
    import tempfile

    import numpy as np

    from oracle import oracle

    out = {}
    cores = os.cpu_count() or 1
    try:  # physical cores = the reference's default thread count (cpu_factory.cc:8-9)
        txt = subprocess.run(["lscpu", "-p=CORE,SOCKET"], capture_output=True, text=True).stdout
        phys_cores = len({l for l in txt.splitlines() if l and not l.startswith("#")})
        if phys_cores > 0:
            cores = phys_cores
    except Exception:
        pass
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, f"ndzip_bench_grid_{os.getpid()}.npy")
    np.save(path, host_grid)
    try:
        # close binding on physical cores was the stable setting on the 2 x 64-core host (tools/cpu_env_probe.sh)
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
        env.pop("OMP_WAIT_POLICY", None)
        r = subprocess.run([sys.executable, "-m", "oracle.timing", path, str(cores), "20"], capture_output=True, text=True, cwd=ROOT, env=env,
                           timeout=300)
        t = json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        os.remove(path)
    c, d = t["compress_GBps_median"], t["decompress_GBps_median"]
    out["cpu_baseline"] = {
        "value": round(2.0 / (1.0 / c + 1.0 / d), 3),
        "unit": "GB/s",
        "cores": cores,
        "kind": "port",
        "sample": f"full {'x'.join(map(str, host_grid.shape))} {host_grid.dtype} grid, median of {t['reps']} reps compress+decompress, OpenMP port "
                  f"of the reference CPU codec (oracle/ndzip_oracle.c), threads pinned to physical cores",
        "compress_GBps": round(c, 3),
        "decompress_GBps": round(d, 3),
        "compress_GBps_best": round(t["compress_GBps_best"], 3),
        "decompress_GBps_best": round(t["decompress_GBps_best"], 3),
        "roundtrip_ok": t["roundtrip_ok"],
    }
    if oracle.have_ref():
        sample = host_grid[: max(16, host_grid.shape[0] // 8)]  # 64 z-planes of the 512^3 grid = 64 MiB
        oracle.ref_compress(sample[:16])
        t0 = time.perf_counter()
        s = oracle.ref_compress(sample)
        t1 = time.perf_counter()
        oracle.ref_decompress(s, sample.dtype, sample.shape)
        t2 = time.perf_counter()
        out["cpu_reference_serial"] = {
            "value": round(2 * sample.nbytes / (t2 - t0) / 1e9, 3),
            "unit": "GB/s",
            "cores": 1,
            "kind": "reference",
            "sample": f"first {sample.shape[0]} z-planes ({sample.nbytes >> 20} MiB), reference serial CPU path compiled from /root/reference (oracle/_ref)",
            "compress_GBps": round(sample.nbytes / (t1 - t0) / 1e9, 3),
            "decompress_GBps": round(sample.nbytes / (t2 - t1) / 1e9, 3),
        }
    if oracle.have_ref() and host_grid.ndim == 3 and host_grid.shape[0] >= 32:
        try:
            out["cpu_reference_slabs"] = cpu_reference_slabs(host_grid, cores)
        except Exception as e:  # a reported extra, never fatal
            out["cpu_reference_slabs"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
    return out


def cpu_reference_slabs(host_grid, cores):
    """The genuine reference serial compressor (oracle/_ref), one z-slab per thread: what a caller of the reference gets from the
    host's cores without the reference's OpenMP back-end (which needs Boost and does not build here).  Every slab is an
    independent ndzip stream of 16 k z-planes; ctypes releases the GIL around the calls; buffers are first touched by the
    thread that uses them."""
    import threading

    import numpy as np

    from oracle import oracle

    planes = host_grid.shape[0]
    threads = max(1, min(cores, planes // 16))
    per = (planes // threads) // 16 * 16
    threads = min(threads, planes // per)
    slabs = [host_grid[i * per: (i + 1) * per] for i in range(threads)]
    tc = [0.0] * threads
    td = [0.0] * threads
    ok = [False] * threads
    # (timeouts + abort: a failing worker must not leave the others, or the benchmark, waiting)
    start = threading.Barrier(threads + 1, timeout=120)
    mid = threading.Barrier(threads + 1, timeout=120)
    end = threading.Barrier(threads + 1, timeout=120)
    wdt = np.uint32 if host_grid.itemsize == 4 else np.uint64

    def work(i):
        try:
            local = np.array(slabs[i], copy=True)                 # first touch on this thread
            oracle.ref_compress(local[:16])                        # warm the code
            start.wait()
            t0 = time.perf_counter()
            s = oracle.ref_compress(local)
            tc[i] = time.perf_counter() - t0
            mid.wait()
            t0 = time.perf_counter()
            back, _ = oracle.ref_decompress(s, local.dtype, local.shape)
            td[i] = time.perf_counter() - t0
            end.wait()
            ok[i] = bool(np.array_equal(back.view(wdt).reshape(-1), local.view(wdt).reshape(-1)))
        except Exception:
            for b in (start, mid, end):
                b.abort()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    mid.wait()
    t1 = time.perf_counter()
    end.wait()
    t2 = time.perf_counter()
    for t in ts:
        t.join()
    nbytes = threads * per * host_grid[0].nbytes
    return {
        "value": round(2 * nbytes / (t2 - t0) / 1e9, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "reference",
        "sample": f"{threads} z-slabs of {per} planes ({nbytes >> 20} MiB in all), each compressed and decompressed by the reference's serial CPU "
                  f"path (oracle/_ref) on its own thread, wall time of the slowest",
        "compress_GBps": round(nbytes / (t1 - t0) / 1e9, 3),
        "decompress_GBps": round(nbytes / (t2 - t1) / 1e9, 3),
        "roundtrip_ok": all(ok),
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import ndzip_amd
    from ndzip_amd.sharded import ShardedCodec
    from ndzip_amd.synth import synth_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with python -m torch.distributed.run --nproc-per-node N bench.py ...")
    # NDZIP_BENCH_SHARE_GPU=1 (self-test on a one-GPU box only): every rank uses cuda:0 and the group is gloo -- RCCL
    # refuses two ranks on one device; same code path otherwise
    share_gpu = world > 1 and os.environ.get("NDZIP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    per_gpu = tuple(int(x) for x in args.shape.split(","))
    dims = len(per_gpu)
    np_dtype = np.dtype(args.dtype)
    t_dtype = torch.float32 if np_dtype == np.float32 else torch.float64
    global_extent = (per_gpu[0] * world,) + per_gpu[1:]
    codec = ShardedCodec(np_dtype, global_extent, rank, world, device)
    shard = codec.shard

    # ---- synthetic input, generated directly in HBM (identical bits on every machine) ------------------------------
    if args.data == "synthetic":
        # the slab's values depend on the GLOBAL coordinates: generate with the global linear offset
        full = None
        n_local = int(np.prod(shard.extent))
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

    def step(ev=None):
        # ev[0..1] bracket the compress launch (descriptor memset + compress kernel [+ border kernel]) on the stream it
        # runs on, ev[2..3] the decompress launch; the offset / header exchange of the N > 1 path lies between them
        codec.compress(local, kernel_events=(ev[0], ev[1]) if ev else None)
        if not args.compress_only:
            if ev:
                ev[2].record()
            codec.decompress(out)
            if ev:
                ev[3].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if not args.compress_only:
        codec.check()

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if not args.compress_only:
        codec.check()

    t_comp = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps * 1e-3   # seconds per launch
    t_decomp = 1e-9 if args.compress_only else sum(e[2].elapsed_time(e[3]) for e in events) / args.steps * 1e-3

    # ---- verification (outside the timed region): round trip is bit-exact; stream hash for the record ---------------
    body_len = int(codec.body_len.cpu()[0]) & 0xFFFFFFFF
    ok = True
    if not args.no_verify and not args.compress_only:
        ok = bool(torch.equal(out.view(torch.int32 if np_dtype == np.float32 else torch.int64),
                              local.view(torch.int32 if np_dtype == np.float32 else torch.int64)))
    stats = torch.tensor([elapsed, t_comp, t_decomp, float(body_len), float(ok)], dtype=torch.float64, device=device)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        mn = stats.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        elapsed, t_comp, t_decomp = float(mx[0]), float(mx[1]), float(mx[2])
        total_body_words = float(sm[3])
        ok = bool(mn[4] > 0.5)
    else:
        total_body_words = float(body_len)

    if rank == 0:
        wb = np_dtype.itemsize
        nhc_total = ndzip_amd.num_hypercubes(global_extent)
        raw_total = raw_bytes_local * world
        stream_bytes_total = (ndzip_amd.header_words(np_dtype, nhc_total) + total_body_words) * wb
        ratio = stream_bytes_total / raw_total
        value = 2 * raw_total * args.steps / elapsed / 1e9
        comp_gbps = raw_total / t_comp / 1e9
        decomp_gbps = raw_total / t_decomp / 1e9
        algo_bytes_per_launch = raw_bytes_local + stream_bytes_total / world   # per GPU: N read + C written
        achieved = algo_bytes_per_launch / t_comp / 1e9
        result = {
            "metric": "compress + decompress GB/s (uncompressed) per GPU; % of HBM3E peak",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if wb == 4 else "u64",
            "data": "synthetic" if args.data == "synthetic" else args.data,
            "config": {
                "workload": f"{dims}D {np_dtype.name} {'x'.join(map(str, global_extent))} synthetic turbulence "
                            f"(Appendix-B generator seed 1, noise_mask {args.noise_mask:#x}{', smooth' if args.smooth else ''}); "
                            f"{world} z-slab(s) of {'x'.join(map(str, per_gpu))}",
                "hypercubes": nhc_total,
                "compression_ratio": round(ratio, 4),
                "step": "compress then decompress, inputs resident in HBM",
                "parallelism": f"hypercube-range sharding x{world}" + (" (RCCL all-gather of offsets + header)" if world > 1 else ""),
            },
            "per_gpu": {
                "compress_GBps": round(comp_gbps / world, 2),
                "decompress_GBps": round(decomp_gbps / world, 2),
                "compress_frac_of_hbm_peak": round(comp_gbps / world / HBM_PEAK_GBPS, 4),
                "decompress_frac_of_hbm_peak": round(decomp_gbps / world / HBM_PEAK_GBPS, 4),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"compress_kernel_db<float,{dims}>" if wb == 4 else f"compress_kernel_wide<unsigned long,{dims}>",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBPS, 4),
                "traffic": None,
                "algorithmic_bytes_per_launch": int(algo_bytes_per_launch),
                "launch_ms": round(t_comp * 1e3, 4),
                "decompress": {
                    "achieved": round(algo_bytes_per_launch / t_decomp / 1e9, 2),
                    "frac": round(algo_bytes_per_launch / t_decomp / 1e9 / HBM_PEAK_GBPS, 4),
                    "launch_ms": round(t_decomp * 1e3, 4),
                },
            },
            "roundtrip_bit_exact": ok,
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                with open(traffic_file) as f:
                    tr = json.load(f)
                key = f"{np_dtype.name}-{'x'.join(map(str, per_gpu))}"
                if key in tr:
                    result["roofline"]["traffic"] = tr[key]["compress_hbm_bytes_per_launch"]
                    result["roofline"]["traffic_source"] = tr[key].get("source")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            try:
                result.update(cpu_baseline(local.cpu().numpy(), dims))
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

    if shard.start0 == 0 and shard.extent == tuple(global_extent):
        return synth_torch(global_extent, t_dtype, seed=1, noise_mask=noise_mask, smooth=smooth, device=device)
    # generate plane blocks of the global grid and keep only this slab
    planes = shard.extent[0]
    rest = tuple(global_extent[1:])
    out = torch.empty(shard.extent, dtype=t_dtype, device=device)
    from ndzip_amd.synth import synth_torch_range

    per_plane = 1
    for x in rest:
        per_plane *= x
    flat = out.view(-1)
    synth_torch_range(global_extent, t_dtype, shard.start0 * per_plane, planes * per_plane, flat, seed=1,
                      noise_mask=noise_mask, smooth=smooth)
    return out


if __name__ == "__main__":
    main()
