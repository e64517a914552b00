"""CPU suite for the multi-GPU path: shard planning and, with two or three gloo processes, the offset
exchange + header gather -- (1) with the ORACLE as the per-rank codec (a test of the host logic: the length all-gather,
ndzip_amd.sharded.base_from_lengths / gather_headers / assemble_stream) and (2) with the real per-rank driver
ndzip_amd.sharded.ShardedCodec, its kernels (compress_split, offset_header_gathered, decompress_split) running on the wave64
functional model of tests/wavesim."""
import os
import socket

import numpy as np
import pytest

from ndzip_amd.sharded import SIDE, assemble_stream, plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle


@pytest.mark.parametrize("extent,world", [((512, 512, 512), 8), ((2048, 1024, 1024), 8), ((1024, 1024, 1024), 8), ((8192, 8192), 8),
                                          ((50, 37, 41), 3), ((70, 200), 2), ((12305,), 4), ((5,), 2), ((16, 16, 16), 4), ((100, 100), 3)])
def test_plan_shards_partitions_hypercubes_and_elements(extent, world):
    shards = plan_shards(extent, world)
    assert len(shards) == world
    assert sum(s.num_hypercubes for s in shards) == oracle.num_hypercubes(extent)
    assert sum(s.border for s in shards) == oracle.border_count(extent)
    assert sum(s.extent[0] for s in shards) == extent[0]
    pos, hc = 0, 0
    side = SIDE[len(extent)]
    for s in shards:
        assert s.start0 == pos and s.hc_begin == hc
        assert s.extent[1:] == tuple(extent[1:])
        if s.rank + 1 < world:
            assert s.extent[0] % side == 0
        assert s.num_hypercubes == oracle.num_hypercubes(s.extent) and s.border == oracle.border_count(s.extent)
        pos += s.extent[0]
        hc = s.hc_end


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, extent, dtype_name, out_dir):
    import torch
    import torch.distributed as dist

    from ndzip_amd.sharded import base_from_lengths, gather_headers

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    wdt = np.uint32 if dtype.itemsize == 4 else np.uint64
    full = synth_numpy(extent, dtype.type, seed=77, noise_mask=0xFF)
    shards = plan_shards(extent, world)
    sh = shards[rank]
    local = np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])
    # per-rank codec with LOCAL offsets (what ndzip_hip_compressor_compress_split produces): header + body(+border)
    stream = oracle.compress(local)
    nhc = sh.num_hypercubes
    hw = (nhc + (1 if wdt == np.uint32 else 2) - 1) // (1 if wdt == np.uint32 else 2)
    header_local = np.frombuffer(stream.tobytes(), dtype=np.uint32)[:nhc].copy()
    body = stream[hw:]
    # the exchange of ShardedCodec.compress: all-gather of one uint32 length (words incl. the local border) per rank ...
    body_len = torch.tensor([len(body)], dtype=torch.int32)
    lens_all = torch.zeros(world, dtype=torch.int32)
    dist.all_gather_into_tensor(lens_all, body_len)
    # ... the base (host restatement of the fused kernel) added to the local entries ...
    base = base_from_lengths(lens_all.numpy().view(np.uint32), [s.border for s in shards], rank)
    hdr = torch.from_numpy(((header_local.astype(np.uint64) + base) & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
    # ... and the all-gather of the header segments
    header_global = gather_headers(hdr, [s.num_hypercubes for s in shards], world)
    total = int(sum(int(x) - s.border for x, s in zip(lens_all.numpy().view(np.uint32), shards)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), header=header_global.numpy().view(np.uint32), body=body.view(wdt), base=int(base), total=int(total))
    dist.destroy_process_group()


@pytest.mark.parametrize("extent,dtype", [((64, 48, 32), np.float32), ((130, 200), np.float64), ((50, 37, 41), np.float32), ((3 * 4096 + 5,), np.float64)])
def test_two_rank_gloo_exchange_reproduces_single_stream(tmp_path, extent, dtype):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, extent, np.dtype(dtype).name, str(tmp_path)), nprocs=world, join=True)
    full = synth_numpy(extent, dtype, seed=77, noise_mask=0xFF)
    want = oracle.compress(full)
    shards = plan_shards(extent, world)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert np.array_equal(parts[0]["header"], parts[1]["header"]), "every rank must hold the same global header"
    assert int(parts[0]["base"]) == 0 and int(parts[1]["base"]) == len(parts[0]["body"]) - shards[0].border
    got = assemble_stream(dtype, extent, parts[0]["header"], [p["body"] for p in parts], [len(p["body"]) for p in parts], shards)
    assert len(got) == len(want) and np.array_equal(got, want)


def _model_rank_main(rank, world, port, extent, dtype_name, out_dir, async_gather):
    """One rank of the REAL per-rank driver (ndzip_amd.sharded.ShardedCodec: compress_split kernel, length all-gather,
    offset_header_gathered kernel, header all-gather, decompress_split kernel) with the kernels running on the wave64
    functional model (tests/wavesim, test infrastructure) and gloo standing in for RCCL."""
    import torch
    import torch.distributed as dist

    from ndzip_amd.sharded import ShardedCodec
    from tests.wavesim import sim

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    wdt = np.uint32 if dtype.itemsize == 4 else np.uint64
    full = synth_numpy(extent, dtype.type, seed=77, noise_mask=0xFF)
    with sim.active():
        codec = ShardedCodec(dtype, extent, rank, world, torch.device("cpu"), async_header_gather=async_gather)
        sh = codec.shard
        local = torch.from_numpy(np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]]))
        out = torch.zeros_like(local)
        for _ in range(3):  # the handle is reused: epochs, ticket reset, pending gathers
            codec.compress(local)
            codec.decompress(out)
        codec.check()
        n = int(codec.body_len.numpy().view(np.uint32)[0])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), header=codec.header_global.numpy().view(np.uint32).copy(),
                 body=codec.body[:n].numpy().view(wdt).copy(), base=int(codec.base32.numpy().view(np.uint32)[0]),
                 roundtrip=bool(np.array_equal(out.numpy().view(wdt), local.numpy().view(wdt))))
    dist.destroy_process_group()


@pytest.mark.parametrize("extent,dtype,world,async_gather", [((64, 48, 32), np.float32, 2, False), ((130, 200), np.float64, 2, True),
                                                           ((50, 37, 41), np.float32, 2, False), ((6 * 4096 + 5,), np.float64, 3, True),
                                                           ((96, 32, 32), np.float32, 3, False),
                                                           # eight ranks, the driver's SCALE shape in small: cfg 4's plan (8 equal slabs, no border) ...
                                                           ((128, 32, 32), np.float32, 8, True),
                                                           # ... and eight unequal slabs of a 2D f64 grid with a border (the last rank takes the tail rows)
                                                           ((64 * 11 + 5, 130), np.float64, 8, False)])
def test_sharded_codec_on_the_model_over_gloo(tmp_path, extent, dtype, world, async_gather):
    import torch.multiprocessing as mp

    from tests.wavesim import build as simbuild

    simbuild.build()  # once, before the ranks race for it
    port = _free_port()
    mp.spawn(_model_rank_main, args=(world, port, extent, np.dtype(dtype).name, str(tmp_path), async_gather), nprocs=world, join=True)
    full = synth_numpy(extent, dtype, seed=77, noise_mask=0xFF)
    want = oracle.compress(full)
    shards = plan_shards(extent, world)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert all(bool(p["roundtrip"]) for p in parts), "a rank's decompress_split did not reproduce its slab"
    for p in parts[1:]:
        assert np.array_equal(parts[0]["header"], p["header"]), "every rank must hold the same global header"
    base = 0
    for p, s in zip(parts, shards):
        assert int(p["base"]) == base
        base += len(p["body"]) - s.border
    got = assemble_stream(dtype, extent, parts[0]["header"], [p["body"] for p in parts], [len(p["body"]) for p in parts], shards)
    assert len(got) == len(want) and np.array_equal(got, want)


@pytest.mark.parametrize("world,mode", [(2, []), (3, ["--async-header-gather"]), (2, ["--overlap-exchange"]), (3, ["--overlap-exchange"]),
                                        (2, ["--native"]), (3, ["--native"]), (3, ["--native", "--overlap-exchange"])],
                         ids=["2-sync", "3-async-header", "2-overlap", "3-overlap", "2-native", "3-native", "3-native-overlap"])
def test_rccl_parity_harness_rehearsed_over_gloo_on_the_model(tmp_path, world, mode):
    """tests/test_hip_sharded_rccl.py's harness -- torch.distributed.run, tests/mp/sharded_rank_main.py, assembly, oracle
    comparison -- with backend gloo and the kernels on the functional model: what runs on the first multi-GPU node is this,
    with `--backend nccl` and one GPU per rank."""
    from tests.test_hip_sharded_rccl import cases_for, check_against_oracle, launch_ranks
    from tests.wavesim import build as simbuild

    simbuild.build()  # once, before the ranks race for it
    cases = cases_for(world)
    launch_ranks(world, tmp_path, cases, "gloo", extra=["--model"] + mode, timeout=500)
    check_against_oracle(world, tmp_path, cases)


def test_cfg4_plan_for_eight_ranks():
    """BASELINE configs[3] (3D float32 2048x1024x1024 over 8 GPUs): 8 slabs of 256 planes = 65 536 hypercubes each, a 256 KiB
    header slice per rank, 2 MiB of header after the all-gather, no border; the per-rank sizes the ShardedCodec buffers take."""
    import ndzip_amd

    extent, world = (2048, 1024, 1024), 8
    shards = plan_shards(extent, world)
    assert [s.extent for s in shards] == [(256, 1024, 1024)] * 8 and [s.start0 for s in shards] == [256 * r for r in range(8)]
    assert all(s.num_hypercubes == 65536 and s.border == 0 for s in shards)
    assert [s.hc_begin for s in shards] == [65536 * r for r in range(8)] and shards[-1].hc_end == 524288
    assert ndzip_amd.header_words(np.float32, 65536) * 4 == 256 << 10 and ndzip_amd.header_words(np.float32, 524288) * 4 == 2 << 20
    # a rank's stream bound: its header slice + 65 536 x 4224 words -- and the whole array's bound is the sum (no rank needs more
    # than a uint32 offset: 8 GiB in, at most 8.25 GiB out = 2.2e9 words < 2^32)
    per_rank = ndzip_amd.compressed_length_bound(np.float32, shards[0].extent)
    assert per_rank == 65536 + 65536 * 4224
    assert ndzip_amd.compressed_length_bound(np.float32, extent) == 8 * per_rank < 2 ** 32
    # cfg5 (3D float64 1024^3 x8, decompress-only): 128 planes = 32 768 hypercubes per rank
    s5 = plan_shards((1024, 1024, 1024), 8)
    assert all(s.extent == (128, 1024, 1024) and s.num_hypercubes == 32768 for s in s5)


def test_plans_the_format_cannot_carry_are_refused_on_the_host():
    """index_type = uint32 (include/ndzip/ndzip.hh:20) counts elements and addresses stream words: eight legal slabs can form a
    global array that is not legal.  ShardedCodec refuses such an extent before anything is allocated; the host restatement of the
    offset kernel refuses lengths that add up past 2^32 - 1 on every rank, not only on the ranks whose own base overflows."""
    from ndzip_amd.sharded import ShardedCodec, base_from_lengths, check_global_extent

    check_global_extent(np.float64, (2048, 1024, 1024))           # bench.py --config 16gib: 2^31 elements, bound 2.2e9 words
    check_global_extent(np.float32, (2048, 1024, 1024))           # cfg 4
    with pytest.raises(ValueError, match="elements"):
        check_global_extent(np.float32, (65536, 65536))           # 2^32 elements; each of 8 slabs (8192 x 65536) is fine by itself
    check_global_extent(np.float32, (8192, 65536))
    with pytest.raises(ValueError, match="32-bit offsets"):
        check_global_extent(np.float32, (65536, 65536 - 64))      # countable, but incompressible data would not be addressable
    with pytest.raises(ValueError):
        ShardedCodec(np.float32, (65536, 65536), 0, 8, "cpu")     # (raises before any library call or allocation)
    borders = [0, 0, 5]
    assert base_from_lengths([0x7FFFFFFF, 0x7FFFFFFF, 6], borders, 2) == 0xFFFFFFFE
    for rank in range(3):  # rank 0's own base is 0 and rank 1's fits: they must refuse all the same
        with pytest.raises(OverflowError):
            base_from_lengths([0xF0000000, 0x20000000, 5], borders, rank)


def _overflow_rank_main(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from ndzip_amd import hip
    from ndzip_amd.sharded import ShardedCodec
    from tests.wavesim import sim

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    extent, dtype = (96, 32, 32), np.float32
    full = synth_numpy(extent, dtype, seed=5, noise_mask=0xFF)
    with sim.active():
        codec = ShardedCodec(dtype, extent, rank, world, torch.device("cpu"))
        sh = codec.shard
        local = torch.from_numpy(np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]]))
        codec.compress(local)
        codec.check()  # a plan that fits: clean
        base_ok = int(codec.base32.numpy().view(np.uint32)[0])
        # every rank's all-gathered lengths forged alike (what eight 4-GiB-class slabs of incompressible data would exchange)
        codec.lens_all.copy_(torch.from_numpy(np.array([0xF0000000, 0x20000000, 7][:world], dtype=np.uint32).view(np.int32)))
        codec.globalise()
        message = ""
        try:
            codec.check()
        except hip.NdzipHipError as e:
            message = str(e)
        # nothing wrapped is ever published: on overflow the entries stay LOCAL and the base 0, a state the rank still decodes its
        # own slab from (the error word says the global stream does not exist)
        codec.compress_local(local)
        local_entries = codec.header_local.numpy().view(np.uint32)[: sh.num_hypercubes].copy()
        codec.globalise()  # (lens_all still holds the forged lengths)
        out = torch.zeros_like(local)
        codec.decompress(out)
        kept_local = bool(np.array_equal(codec.header_local.numpy().view(np.uint32)[: sh.num_hypercubes], local_entries)) and \
            int(codec.base32.numpy().view(np.uint32)[0]) == 0 and bool(np.array_equal(out.numpy().view(np.uint32), local.numpy().view(np.uint32)))
        with pytest.raises(hip.NdzipHipError, match="32-bit offsets"):
            codec.check()
        codec.compress(local)  # the error word was cleared by check(): the handle works again
        codec.check()
        again = int(codec.base32.numpy().view(np.uint32)[0])
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{base_ok}\n{again}\n{message}\n{kept_local}\n")
    dist.destroy_process_group()


def test_overflowing_offsets_raise_on_every_rank_on_the_model_over_gloo(tmp_path):
    """offset_header_gathered_kernel sums the all-gathered lengths of ALL shards in 64 bits: a plan whose hypercube runs exceed the
    format's 32-bit offsets sets the error-word bit on every rank (rank 0, whose own base is 0, included), ShardedCodec.check()
    raises there, and the handle is usable afterwards."""
    import torch.multiprocessing as mp

    from tests.wavesim import build as simbuild

    simbuild.build()
    world, port = 3, _free_port()
    mp.spawn(_overflow_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        base_ok, again, message, kept_local = (tmp_path / f"rank{r}.txt").read_text().split("\n")[:4]
        assert base_ok == again and (r > 0) == (int(base_ok) > 0) and kept_local == "True"
        assert "32-bit offsets" in message and "0x4" in message, (r, message)
