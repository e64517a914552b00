"""One rank of the N-rank parity run of ndzip_amd.sharded.ShardedCodec, launched by torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tests/mp/sharded_rank_main.py --backend nccl --out DIR --case float32:96,64,48 [--case ...]

--backend nccl   one GPU per rank (cuda:LOCAL_RANK), RCCL collectives on device tensors: tests/test_hip_sharded_rccl.py (-m gpu)
--backend gloo --model   host tensors, the kernels' functional model (tests/wavesim): the rehearsal of the very same script in
                 the GPU-less container, tests/test_sharded_cpu.py

Per case every rank compresses its slab three times on one ShardedCodec (handle reuse: descriptor epochs, ticket reset, header
buffers), decompresses it, and writes rank<r>_case<i>.npz = its header_global, body, base and round-trip verdict; the test
process assembles the stream from those files and compares it with the oracle's."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 47


def case_data(case):
    """(dtype, extent, full array) of a `dtype:e0,e1,...` case (the test process builds the same array for the oracle)."""
    from ndzip_amd.synth import synth_numpy

    name, ext = case.split(":")
    dtype = np.dtype(name).type
    extent = tuple(int(x) for x in ext.split(","))
    return dtype, extent, synth_numpy(extent, dtype, seed=SEED, noise_mask=0xFF)


def main():
    import contextlib

    import torch
    import torch.distributed as dist

    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["nccl", "gloo"], required=True)
    ap.add_argument("--model", action="store_true", help="kernels on the functional model, host tensors (CPU rehearsal)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--case", action="append", required=True)
    ap.add_argument("--async-header-gather", action="store_true")
    ap.add_argument("--overlap-exchange", action="store_true", help="decode from local offsets while the exchange runs behind it")
    ap.add_argument("--native", action="store_true", help="the C++ host (libndzip_hip_rccl.so, include/ndzip_hip_sharded.h: its own ncclComm_t, "
                                                          "RCCL called from C++) instead of the torch.distributed driver")
    args = ap.parse_args()
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))

    from ndzip_amd.sharded import ShardedCodec

    if args.model:
        from tests.wavesim import sim

        device = torch.device("cpu")
        scope = sim.active()
    else:
        assert torch.cuda.is_available() and torch.cuda.device_count() > local_rank, "one visible GPU per rank"
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        scope = contextlib.nullcontext()
    kw = {"device_id": device} if args.backend == "nccl" else {}
    dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
    with scope:
        for i, case in enumerate(args.case):
            dtype, extent, full = case_data(case)
            wdt = np.uint32 if np.dtype(dtype).itemsize == 4 else np.uint64
            if args.native:
                native_case(args, i, dtype, extent, full, wdt, rank, world, device)
                dist.barrier()
                continue
            codec = ShardedCodec(dtype, extent, rank, world, device, async_header_gather=args.async_header_gather,
                                 overlap_exchange=args.overlap_exchange)
            sh = codec.shard
            slab = torch.from_numpy(np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])).to(device)
            out = torch.zeros_like(slab)
            for _ in range(3):
                codec.compress(slab)
                codec.decompress(out)
            codec.check()
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            n = int(codec.body_len.cpu().numpy().view(np.uint32)[0])
            np.savez(os.path.join(args.out, f"rank{rank}_case{i}.npz"),
                     header=codec.header_global.cpu().numpy().view(np.uint32).copy(),
                     body=codec.body[:n].cpu().numpy().view(wdt).copy(),
                     base=int(codec.base32.cpu().numpy().view(np.uint32)[0]),
                     roundtrip=bool(np.array_equal(out.cpu().numpy().reshape(-1).view(wdt), slab.cpu().numpy().reshape(-1).view(wdt))))
            dist.barrier()
    dist.destroy_process_group()


def native_case(args, i, dtype, extent, full, wdt, rank, world, device):
    """The same case through ndzip_amd.sharded_native.NativeShardedCodec.  What the test process reads is taken from the rank's
    pieces of the single stream (ndzip_hip_sharded_write_stream into a host buffer of the whole stream's size) and its layout."""
    import ctypes as C

    import torch

    from ndzip_amd import sharded_native

    kw = {}
    if args.model:  # sharded.cc compiled against the functional model; gloo carries the one-function exchange
        from tests.test_sharded_native_cpu import _gloo_table
        from tests.wavesim import build as simbuild

        sharded_native._lib = sharded_native._bind(C.CDLL(simbuild.build_sharded()), rccl=False)
        table, _ = _gloo_table(world)
        kw["collectives"] = table
    codec = sharded_native.NativeShardedCodec(dtype, extent, rank, world, device, overlap_exchange=args.overlap_exchange, **kw)
    sh = codec.shard
    slab = torch.from_numpy(np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])).to(device)
    out = torch.zeros_like(slab)
    for _ in range(3):
        codec.compress(slab)
        codec.decompress(out)
    codec.check()
    lay = codec.stream_layout()
    stream = np.zeros(int(lay.stream_words), dtype=wdt)
    codec.write_stream(stream, with_header=True)
    nhc = sum(s.num_hypercubes for s in codec.shards)
    r0, r1, b0, b1 = int(lay.runs_offset_words), int(lay.runs_offset_words + lay.runs_words), int(lay.border_offset_words), int(lay.border_offset_words + lay.border_words)
    np.savez(os.path.join(args.out, f"rank{rank}_case{i}.npz"),
             header=stream.view(np.uint32)[:nhc].copy(), body=np.concatenate([stream[r0:r1], stream[b0:b1]]),
             base=r0 - int(lay.header_words),
             roundtrip=bool(np.array_equal(out.cpu().numpy().reshape(-1).view(wdt), slab.cpu().numpy().reshape(-1).view(wdt))))
    codec.close()


if __name__ == "__main__":
    main()
