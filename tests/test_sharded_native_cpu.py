"""CPU suite for the C++ host of the multi-GPU path (include/ndzip_hip_sharded.h, ndzip_amd/csrc/sharded.cc + sharded_rccl.cc ->
ndzip_amd/libndzip_hip_rccl.so): the library loads and exports every declared symbol, its plan is the Python plan, it refuses what
the format cannot carry -- and, with the SAME sharded.cc compiled against the kernels' functional model (tests/wavesim) and a
gloo-backed collectives table standing in for RCCL, 2 / 3 / 8 ranks reproduce the oracle's single stream byte for byte and decode
their slabs back from it."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

from ndzip_amd import hip, sharded_native
from ndzip_amd.sharded import plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = sharded_native.lib()
    for name in sharded_native.EXPORTED_SYMBOLS:
        getattr(L, name)
    assert L.ndzip_hip_sharded_abi_version() == sharded_native.ABI_VERSION
    # ... and the header declares exactly these
    with open(os.path.join(ROOT, "include", "ndzip_hip_sharded.h")) as f:
        declared = set(re.findall(r"NDZIP_HIP_API\s+[\w\s\*]+?\b(ndzip_hip_\w+)\s*\(", f.read()))
    assert declared == set(sharded_native.EXPORTED_SYMBOLS)
    # the product library stays free of RCCL; the sharded one carries it
    import subprocess

    needed = lambda p: subprocess.run(["readelf", "-d", p], capture_output=True, text=True).stdout
    assert "librccl" not in needed(hip.LIB_PATH) and "librccl.so.1" in needed(sharded_native.LIB_PATH) and "libndzip_hip.so" in needed(sharded_native.LIB_PATH)


@pytest.mark.parametrize("extent,world", [((512, 512, 512), 8), ((2048, 1024, 1024), 8), ((1024, 1024, 1024), 8), ((8192, 8192), 8), ((1 << 24,), 8),
                                          ((50, 37, 41), 3), ((70, 200), 2), ((12305,), 4), ((5,), 2), ((16, 16, 16), 4), ((100, 100), 3),
                                          ((64 * 11 + 5, 130), 8), ((40, 5, 100), 3), ((2048, 1024, 1024), 1)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_plan_is_the_python_plan(extent, world, dtype):
    want = plan_shards(extent, world)
    for r in range(world):
        sh = sharded_native.plan(dtype, extent, r, world)
        assert (sh.rank, sh.world, sh.start0, tuple(sh.extent[: len(extent)]), sh.hc_begin, sh.hc_end, sh.border_elements) == \
            (r, world, want[r].start0, want[r].extent, want[r].hc_begin, want[r].hc_end, want[r].border)
        assert sh.body_capacity_words == hip.compressed_length_bound(dtype, want[r].extent) - hip.header_words(dtype, want[r].num_hypercubes)


def test_bad_plans_and_extents_the_format_cannot_carry_are_refused_on_the_host():
    with pytest.raises(hip.NdzipHipError, match="outside a plan"):
        sharded_native.plan(np.float32, (64, 64), 2, 2)
    with pytest.raises(hip.NdzipHipError, match="dimensionality"):
        sharded_native.plan(np.float32, (4, 4, 4, 4), 0, 1)
    L = sharded_native.lib()

    @sharded_native.ALL_GATHER_U32
    def never(ctx, send, recv, count, stream):
        return 1

    table = sharded_native.Collectives(None, never, sharded_native.ERROR_STRING())
    for dtype, extent, what in ((np.float64, (4096, 1024, 1024), "2\\^32 - 1 elements"),       # 2^32 elements
                                (np.float32, (4095, 1024, 1024), "32-bit offsets"),            # legal count, runs can exceed the offsets
                                (np.float64, (65536, 65536 - 64), "32-bit offsets"),
                                (np.float32, (3968, 1024, 1024 + 15), "uint32 stream length")):  # runs fit, runs + border + header do not
        h = C.c_void_p()
        st = L.ndzip_hip_sharded_create_with_collectives(hip._dtype_code(dtype), len(extent), sharded_native._ext(extent), 0, 8, C.byref(table), None, C.byref(h))
        assert st == -7 and not h.value, (extent, st)  # NDZIP_HIP_ERR_LIMIT, before any device is touched
        assert re.search(what, L.ndzip_hip_sharded_last_error().decode()), L.ndzip_hip_sharded_last_error()
    # the 16 GiB strong-scaling grid of bench.py is inside all three limits (no GPU here: the next stop is the device)
    h = C.c_void_p()
    st = L.ndzip_hip_sharded_create_with_collectives(1, 3, sharded_native._ext((2048, 1024, 1024)), 0, 8, C.byref(table), None, C.byref(h))
    assert st in (-5, 0), L.ndzip_hip_sharded_last_error()  # NDZIP_HIP_ERR_NO_DEVICE here
    if st == 0:
        L.ndzip_hip_sharded_destroy(h)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_table(world):
    """The one-function exchange over gloo: on the model "device" pointers are host pointers."""
    import torch
    import torch.distributed as dist

    calls = []

    @sharded_native.ALL_GATHER_U32
    def all_gather_u32(ctx, send, recv, count, stream):
        try:
            s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_int32)), shape=(count,)))
            r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_int32)), shape=(world * count,)))
            dist.all_gather_into_tensor(r, s.clone())
            calls.append(count)
            return 0
        except Exception as e:  # (no exception crosses the C ABI)
            print("all_gather_u32 failed:", e)
            return 1

    @sharded_native.ERROR_STRING
    def error_string(ctx, code):
        return b"gloo all-gather failed"

    return sharded_native.Collectives(None, all_gather_u32, error_string), calls


def _native_rank_main(rank, world, port, extent, dtype_name, out_dir):
    import torch
    import torch.distributed as dist

    from ndzip_amd.sharded import ShardedCodec
    from tests.wavesim import build as simbuild
    from tests.wavesim import sim

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    wdt = np.uint32 if dtype.itemsize == 4 else np.uint64
    full = synth_numpy(extent, dtype.type, seed=77, noise_mask=0xFF)
    want = oracle.compress(full)
    sharded_native._lib = sharded_native._bind(C.CDLL(simbuild.build_sharded()), rccl=False)  # sharded.cc on the model
    table, calls = _gloo_table(world)
    result = {}
    with sim.active():
        codec = sharded_native.NativeShardedCodec(dtype, extent, rank, world, torch.device("cpu"), collectives=table)
        sh = codec.shard
        local = np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])
        out = np.zeros_like(local)
        for _ in range(2):  # the handle is reused
            codec.compress(local.ctypes.data)
            codec.decompress(out.ctypes.data)
        codec.check()
        result["roundtrip"] = bool(np.array_equal(out.view(wdt), local.view(wdt)))
        result["gathers"] = calls[-2:] if world > 1 else calls
        lay = codec.stream_layout()
        result["stream_words"] = int(lay.stream_words)
        # every rank copies its pieces into ONE shared mapping of the output file; rank 0 adds the header
        path = os.path.join(out_dir, "stream.bin")
        if rank == 0:
            np.zeros(lay.stream_words, dtype=wdt).tofile(path)
        dist.barrier()
        mm = np.memmap(path, dtype=wdt, mode="r+", shape=(int(lay.stream_words),))
        codec.write_stream(mm, with_header=rank == 0)
        mm.flush()
        # the global header every rank holds == the oracle's
        hp, n, _, _, base_p = codec.pointers()
        hdr = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint32)), shape=(max(1, n),))[:n].copy()
        result["header_ok"] = bool(np.array_equal(hdr, want.view(np.uint32)[:n]))
        result["base"] = int(np.ctypeslib.as_array(C.cast(base_p, C.POINTER(C.c_uint32)), shape=(1,))[0])
        # the torch.distributed driver (ndzip_amd.sharded.ShardedCodec) arrives at the same header and base
        ref = ShardedCodec(dtype, extent, rank, world, torch.device("cpu"))
        ref.compress(torch.from_numpy(local))
        ref.check()
        result["same_as_python_driver"] = bool(np.array_equal(ref.header_global.numpy().view(np.uint32)[:n], hdr)) and \
            int(ref.base32.numpy().view(np.uint32)[0]) == result["base"]
        # the way back, on a fresh handle: this rank's pieces out of the ORACLE's stream, decoded without any collective
        back = sharded_native.NativeShardedCodec(dtype, extent, rank, world, torch.device("cpu"), collectives=table)
        n_calls = len(calls)
        back.load(want)
        out2 = np.zeros_like(local)
        back.decompress(out2.ctypes.data)
        back.check()
        lay2 = back.stream_layout()
        result["load_roundtrip"] = bool(np.array_equal(out2.view(wdt), local.view(wdt))) and len(calls) == n_calls
        result["load_layout_same"] = all(getattr(lay, f) == getattr(lay2, f) for f, _ in lay._fields_)
        # misuse: a second exchange without a compress_local is refused, not applied twice
        with pytest.raises(hip.NdzipHipError, match="exchange without"):
            sharded_native._check(sharded_native.lib().ndzip_hip_sharded_exchange(codec._h))
        codec.close()
        back.close()
    dist.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **result)
    dist.destroy_process_group()


@pytest.mark.parametrize("extent,dtype,world", [((64, 48, 32), np.float32, 2), ((130, 200), np.float64, 2), ((50, 37, 41), np.float32, 2),
                                                ((6 * 4096 + 5,), np.float64, 3), ((96, 32, 32), np.float32, 3),
                                                ((128, 32, 32), np.float32, 8),            # cfg 4's plan in small: 8 equal slabs, no border
                                                ((64 * 11 + 5, 130), np.float64, 8),       # 8 unequal slabs with a border
                                                ((10, 70), np.float32, 2),                 # no hypercube at all: the stream is one border
                                                ((64, 64), np.float64, 1)])
def test_native_host_reproduces_the_single_stream_on_the_model_over_gloo(tmp_path, extent, dtype, world):
    import torch.multiprocessing as mp

    from tests.wavesim import build as simbuild

    simbuild.build_sharded()  # once, before the ranks race for it
    mp.spawn(_native_rank_main, args=(world, _free_port(), extent, np.dtype(dtype).name, str(tmp_path)), nprocs=world, join=True)
    want = oracle.compress(synth_numpy(extent, dtype, seed=77, noise_mask=0xFF))
    got = np.fromfile(tmp_path / "stream.bin", dtype=want.dtype)
    assert len(got) == len(want) and np.array_equal(got, want), "the ranks' pieces do not add up to the reference stream"
    shards = plan_shards(extent, world)
    nmax = max(s.num_hypercubes for s in shards)
    for r in range(world):
        p = np.load(tmp_path / f"rank{r}.npz")
        for key in ("roundtrip", "header_ok", "same_as_python_driver", "load_roundtrip", "load_layout_same"):
            assert bool(p[key]), (r, key)
        assert int(p["stream_words"]) == len(want)
        # exactly two collectives per compress: one uint32 per rank, then the longest header segment (none when nobody owns a hypercube)
        assert list(p["gathers"]) == ([] if world == 1 else [1, nmax] if nmax else [1, 1])


def test_cpp_threads_host_on_the_model(tmp_path):
    """tests/cpp/sharded_threads.cc -- every rank a thread of ONE C++ program, the exchange a caller-supplied table, no Python in the
    loop -- compiled against the functional model: 2 / 3 / 4 / 8 ranks write the oracle's single stream and decode their slabs back
    from it.  The same program runs on the GPU box in tests/test_hip_sharded_native.py."""
    import subprocess

    from tests.test_hip_sharded_native import THREAD_CASES, THREADS_SRC, run_threads_host
    from tests.wavesim import build as simbuild

    lib = simbuild.build_sharded(variant="")
    here = os.path.dirname(lib)
    exe = str(tmp_path / "sharded_threads_model")
    r = subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-parameter",
                        "-I", here, "-I", os.path.join(ROOT, "include"), THREADS_SRC, "-o", exe, "-L" + here, "-l:" + os.path.basename(lib),
                        "-l:libndzip_hip_wavesim.so", "-Wl,-rpath," + here], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, WAVESIM_CUS="2", WAVESIM_BLOCKS_PER_CU="2")
    env.pop("WAVESIM_VARIANT", None)
    for dtype, extent, world in THREAD_CASES[:5]:
        run_threads_host(tmp_path, exe, dtype, extent, world, env=env)


def test_misuse_and_corrupt_streams_are_refused_on_the_model(monkeypatch):
    """The handle's state rules and the host-side validation of ndzip_hip_sharded_load, on one shard of the model: decode before
    anything was compressed or loaded, layout / write_stream while the exchange is still due, a stream buffer that is too small, a
    truncated stream, a header entry out of order, a stream of another extent -- every one an error code with a message, none a
    wild read."""
    from tests.wavesim import build as simbuild
    from tests.wavesim import sim

    monkeypatch.setattr(sharded_native, "_lib", sharded_native._bind(C.CDLL(simbuild.build_sharded(variant="")), rccl=False))
    L = sharded_native.lib()

    @sharded_native.ALL_GATHER_U32
    def never(ctx, send, recv, count, stream):
        return 1

    table = sharded_native.Collectives(None, never, sharded_native.ERROR_STRING())
    extent, dtype = (130, 200), np.float64
    data = synth_numpy(extent, dtype, seed=3, noise_mask=0xFF)
    want = oracle.compress(data)
    import torch

    with sim.active():
        codec = sharded_native.NativeShardedCodec(dtype, extent, 0, 1, torch.device("cpu"), collectives=table)
        out = np.zeros_like(data)
        with pytest.raises(hip.NdzipHipError, match="nothing to decode"):
            codec.decompress(out.ctypes.data)
        with pytest.raises(hip.NdzipHipError, match="no stream"):
            codec.stream_layout()
        sharded_native._check(L.ndzip_hip_sharded_compress_local(codec._h, data.ctypes.data))
        with pytest.raises(hip.NdzipHipError, match="still local"):
            codec.stream_layout()
        with pytest.raises(hip.NdzipHipError, match="still local"):
            codec.write_stream(np.zeros(len(want), dtype=want.dtype), with_header=True)
        codec.decompress(out.ctypes.data)  # (allowed: the slab decodes from its local offsets)
        assert np.array_equal(out.view(np.uint64), data.view(np.uint64))
        sharded_native._check(L.ndzip_hip_sharded_exchange(codec._h))
        with pytest.raises(hip.NdzipHipError) as e:
            codec.write_stream(np.zeros(len(want) - 1, dtype=want.dtype), with_header=True)
        assert e.value.status == -3 and "the buffer" in str(e.value)  # NDZIP_HIP_ERR_CAPACITY
        got = np.zeros(len(want), dtype=want.dtype)
        codec.write_stream(got, with_header=True)
        assert np.array_equal(got, want)
        # the way back refuses what is not a stream of THIS extent
        back = sharded_native.NativeShardedCodec(dtype, extent, 0, 1, torch.device("cpu"), collectives=table)
        with pytest.raises(hip.NdzipHipError):
            back.load(want[: len(want) - 5])                      # truncated
        bad = want.copy()
        bad.view(np.uint32)[1] = bad.view(np.uint32)[0]            # second entry does not follow the first by a hypercube's length
        with pytest.raises(hip.NdzipHipError):
            back.load(bad)
        other = oracle.compress(synth_numpy((70, 200), dtype, seed=3, noise_mask=0xFF))
        with pytest.raises(hip.NdzipHipError):
            back.load(other)                                       # fewer hypercubes than this extent's header needs
        with pytest.raises(hip.NdzipHipError, match="nothing to decode"):
            back.decompress(out.ctypes.data)                       # (a refused load leaves the handle empty)
        back.load(want)
        out[:] = 0
        back.decompress(out.ctypes.data)
        back.check()
        assert np.array_equal(out.view(np.uint64), data.view(np.uint64))
        codec.close()
        back.close()


ADAPTOR_SRC = os.path.join(ROOT, "tests", "cpp", "sharded_adaptor_roundtrip.cc")
ADAPTOR_CASES = [(np.float32, (50, 37, 41)), (np.float64, (130, 200)), (np.float64, (3 * 4096 + 5,)), (np.float32, (10, 70))]


def run_adaptor_program(tmp_path, exe, env=None):
    import subprocess

    for dtype, extent in ADAPTOR_CASES:
        data = synth_numpy(extent, dtype, seed=9, noise_mask=0xFF)
        (tmp_path / "in.bin").write_bytes(data.tobytes())
        (tmp_path / "ref.bin").write_bytes(oracle.compress(data).tobytes())
        r = subprocess.run([exe, "f32" if np.dtype(dtype).itemsize == 4 else "f64", ",".join(map(str, extent)), str(tmp_path / "in.bin"), str(tmp_path / "ref.bin")],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "round trip ok" in r.stdout, (extent, r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_cpp_adaptor_of_the_sharded_path_on_the_model(tmp_path):
    """include/ndzip_hip_sharded.hh (hip_sharded_codec<T>, the <T, Dims> spelling, exceptions with the C ABI's message) through
    tests/cpp/sharded_adaptor_roundtrip.cc on the functional model: stream == the oracle's, both ways back -- and the header compiles
    against the real HIP headers and, where the reference tree exists, against the reference's own ndzip.hh (its extent type)."""
    import subprocess

    from tests.wavesim import build as simbuild

    lib = simbuild.build_sharded(variant="")
    here = os.path.dirname(lib)
    exe = str(tmp_path / "sharded_adaptor_model")
    r = subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-parameter",
                        "-I", here, "-I", os.path.join(ROOT, "include"), ADAPTOR_SRC, "-o", exe, "-L" + here, "-l:" + os.path.basename(lib),
                        "-l:libndzip_hip_wavesim.so", "-Wl,-rpath," + here], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, WAVESIM_CUS="2", WAVESIM_BLOCKS_PER_CU="2")
    env.pop("WAVESIM_VARIANT", None)
    run_adaptor_program(tmp_path, exe, env)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    base = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include")]
    r = subprocess.run(base + [ADAPTOR_SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    if os.path.isdir("/root/reference/include/ndzip"):
        r = subprocess.run(base[:4] + base[5:] + ["-DNDZIP_HIP_WITH_REFERENCE_HEADERS", "-I/root/reference/include", ADAPTOR_SRC], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
