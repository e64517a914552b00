"""The C++ host of the multi-GPU path (include/ndzip_hip_sharded.h -> ndzip_amd/libndzip_hip_rccl.so) on hardware: the example
program tests/cpp/sharded_host.cc -- one process per GPU, its own ncclComm_t, RCCL called from C++ -- must write the oracle's
single stream for the whole array and decode every slab back from it; NativeShardedCodec (the ctypes binding) on one GPU must do
the same from Python.  With one visible GPU the plan has one shard (no communicator); with N >= 2 the program runs N ranks over
RCCL (the N-rank Python-driven run of the same library is tests/test_hip_sharded_rccl.py, mode native-cpp-host).
The CPU suite compiles and links the program (here) and runs the library's transport-independent half on the kernels' functional
model over gloo (tests/test_sharded_native_cpu.py)."""
import os
import subprocess

import numpy as np
import pytest

from ndzip_amd.synth import synth_numpy
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sharded_host.cc")
LIBDIR = os.path.join(ROOT, "ndzip_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


THREADS_SRC = os.path.join(ROOT, "tests", "cpp", "sharded_threads.cc")


def build_host(exe, src=SRC):
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
                        "-I" + os.path.join(ROOT, "include"), src, "-o", str(exe), "-L" + LIBDIR, "-lndzip_hip_rccl", "-lndzip_hip",
                        "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + os.path.join(ROCM, "lib")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(exe)


def test_example_host_compiles_and_links_against_the_sharded_library(tmp_path):
    exe = build_host(tmp_path / "sharded_host")
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libndzip_hip_rccl.so" in needed  # (which in turn needs libndzip_hip.so: the kernels)
    # (rccl.h is not needed by a host that uses the bootstrap helpers: the program includes only the C ABI and the HIP runtime API)
    with open(SRC) as f:
        assert "rccl.h" not in f.read()


def test_threads_host_compiles_and_links_against_the_sharded_library(tmp_path):
    build_host(tmp_path / "sharded_threads", THREADS_SRC)


def _gpus():
    try:
        import torch

        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def run_host(tmp_path, exe, dtype, extent, world):
    data = synth_numpy(extent, dtype, seed=61, noise_mask=0xFF)
    (tmp_path / "in.bin").write_bytes(data.tobytes())
    out = tmp_path / "stream.bin"
    if out.exists():
        out.unlink()
    idf = tmp_path / "nccl_id"
    if idf.exists():
        idf.unlink()
    base = [exe, "--world", str(world), "--id-file", str(idf), "--dtype", "f32" if np.dtype(dtype).itemsize == 4 else "f64",
            "--extent", ",".join(map(str, extent)), "--in", str(tmp_path / "in.bin"), "--out", str(out)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen(base + ["--rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, start_new_session=True)
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=300)
            logs.append(o)
    except subprocess.TimeoutExpired:
        import signal

        for p in procs:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        pytest.fail(f"{world} ranks of sharded_host did not finish\n" + "\n".join(logs)[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    want = oracle.compress(data)
    got = np.fromfile(out, dtype=want.dtype)
    assert len(got) == len(want) and np.array_equal(got, want), "the ranks' pieces do not add up to the reference stream"
    assert all(": ok" in l for l in logs)


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.parametrize("dtype,extent", [(np.float32, (70, 50, 36)), (np.float64, (130, 200)), (np.float32, (3 * 4096 + 17,)), (np.float64, (40, 33, 18)),
                                          (np.float32, (10, 70))])
def test_example_host_single_gpu(tmp_path, dtype, extent):
    run_host(tmp_path, build_host(tmp_path / "sharded_host"), dtype, extent, 1)


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.skipif(_gpus() < 2, reason="needs at least 2 GPUs (one process per GPU, RCCL called from C++)")
@pytest.mark.parametrize("dtype,extent_of", [(np.float32, lambda w: (16 * w, 32, 48)), (np.float32, lambda w: (16 * (w + 1) + 5, 37, 41)),
                                             (np.float64, lambda w: (64 * (w + 1), 200)), (np.float64, lambda w: (4096 * (2 * w + 1) + 77,))],
                         ids=["f32-3d-equal", "f32-3d-unequal-border", "f64-2d", "f64-1d"])
def test_example_host_all_gpus_over_rccl(tmp_path, dtype, extent_of):
    world = _gpus()
    run_host(tmp_path, build_host(tmp_path / "sharded_host"), dtype, extent_of(world), world)


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.parametrize("dtype,extent", [(np.float32, (70, 50, 36)), (np.float64, (130, 200)), (np.float64, (4096 * 3 + 5,))])
def test_native_codec_from_python_single_gpu(dtype, extent):
    """NativeShardedCodec with one shard on cuda:0: stream == oracle, decode of the resident stream and of a loaded one."""
    import torch

    from ndzip_amd.sharded_native import NativeShardedCodec

    wdt = np.uint32 if np.dtype(dtype).itemsize == 4 else np.uint64
    data = synth_numpy(extent, dtype, seed=62, noise_mask=0xFF)
    want = oracle.compress(data)
    dev = torch.device("cuda", 0)
    codec = NativeShardedCodec(dtype, extent, 0, 1, dev)
    slab = torch.from_numpy(data).to(dev)
    out = torch.zeros_like(slab)
    for _ in range(2):
        codec.compress(slab)
        codec.decompress(out)
    codec.check()
    lay = codec.stream_layout()
    assert lay.stream_words == len(want)
    got = np.zeros(len(want), dtype=wdt)
    codec.write_stream(got, with_header=True)
    assert np.array_equal(got, want)
    assert np.array_equal(out.cpu().numpy().view(wdt), data.view(wdt))
    back = NativeShardedCodec(dtype, extent, 0, 1, dev)
    back.load(want)
    out.zero_()
    back.decompress(out)
    back.check()
    assert np.array_equal(out.cpu().numpy().view(wdt), data.view(wdt))
    codec.close()
    back.close()


THREAD_CASES = [(np.float32, (50, 37, 41), 3), (np.float64, (64 * 5 + 5, 130), 4), (np.float32, (6 * 4096 + 5,), 3), (np.float64, (32, 16, 48), 2),
                (np.float32, (10, 70), 2), (np.float32, (128, 64, 64), 8)]


def run_threads_host(tmp_path, exe, dtype, extent, world, env=None):
    data = synth_numpy(extent, dtype, seed=9, noise_mask=0xFF)
    (tmp_path / "in.bin").write_bytes(data.tobytes())
    out = tmp_path / "stream.bin"
    r = subprocess.run([exe, "--world", str(world), "--dtype", "f32" if np.dtype(dtype).itemsize == 4 else "f64", "--extent", ",".join(map(str, extent)),
                        "--in", str(tmp_path / "in.bin"), "--out", str(out)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    want = oracle.compress(data)
    got = np.fromfile(out, dtype=want.dtype)
    assert len(got) == len(want) and np.array_equal(got, want), "the ranks' pieces do not add up to the reference stream"
    assert r.stdout.count(": ok") == world


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.parametrize("dtype,extent,world", THREAD_CASES)
def test_threads_host_plays_every_rank_on_one_gpu(tmp_path, dtype, extent, world):
    """The world > 1 path of the C++ host (length gather, offset kernel, padded header gather, compaction, layout, load) on a box with
    ONE GPU: every rank a thread of tests/cpp/sharded_threads.cc, the exchange a rendezvous + device-to-device copies."""
    run_threads_host(tmp_path, build_host(tmp_path / "sharded_threads", THREADS_SRC), dtype, extent, world)


@pytest.mark.gpu
@pytest.mark.hardware_only
def test_cpp_adaptor_of_the_sharded_path_on_gpu(tmp_path):
    """include/ndzip_hip_sharded.hh through tests/cpp/sharded_adaptor_roundtrip.cc against the real library (one shard)."""
    from tests.test_sharded_native_cpu import ADAPTOR_SRC, run_adaptor_program

    run_adaptor_program(tmp_path, build_host(tmp_path / "sharded_adaptor", ADAPTOR_SRC))


def _run_tool(tmp_path, dtype, extent, extra, expect):
    """ndzip_amd/ndzip-hip-sharded on the GPU: the stream file == the oracle's stream of the whole array; a reference stream decodes
    back to the array."""
    from ndzip_amd import build

    exe = build.SHARDED_CLI_OUT
    assert os.path.exists(exe), "ndzip_amd/ndzip-hip-sharded is missing: python -m ndzip_amd.build"
    data = synth_numpy(extent, dtype, seed=11, noise_mask=0xFF)
    want = oracle.compress(data)
    src, ndz, ref, back = tmp_path / "in.bin", tmp_path / "out.ndz", tmp_path / "ref.ndz", tmp_path / "back.bin"
    data.tofile(src)
    want.tofile(ref)
    t = ["-t", "float" if np.dtype(dtype).itemsize == 4 else "double"]
    size = [str(x) for x in extent]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "-n", *size, *t, "-i", str(src), "-o", str(ndz), "--repeat", "3", *extra], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and expect in r.stderr, r.stderr[-2000:]
    got = np.fromfile(ndz, dtype=want.dtype)
    assert len(got) == len(want) and np.array_equal(got, want), "the ranks' pieces do not add up to the reference stream"
    r = subprocess.run([exe, "-d", "-n", *size, *t, "-i", str(ref), "-o", str(back), *extra], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.fromfile(back, dtype=want.dtype), data.reshape(-1).view(want.dtype))


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.parametrize("dtype,extent,ranks", [(np.float32, (70, 50, 36), 1), (np.float32, (96, 64, 48), 3), (np.float64, (64 * 5 + 5, 130), 4),
                                                (np.float64, (4096 * 6 + 5,), 2), (np.float32, (256, 256, 256), 4)])
def test_file_tool_several_ranks_on_one_gpu(tmp_path, dtype, extent, ranks):
    """--devices 1: the ranks share GPU 0 (in-process exchange, device work one rank at a time)."""
    _run_tool(tmp_path, dtype, extent, ["--ranks", str(ranks), "--devices", "1"], f"{ranks} rank(s) on 1 GPU(s), exchange {'none' if ranks == 1 else 'local'}")


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.skipif(_gpus() < 2, reason="needs at least 2 GPUs (one rank per GPU, RCCL)")
@pytest.mark.parametrize("exchange", ["rccl", "local"])
def test_file_tool_one_rank_per_gpu(tmp_path, exchange):
    world = _gpus()
    _run_tool(tmp_path, np.float32, (16 * (world + 1) + 5, 64, 48), ["--exchange", exchange], f"{world} rank(s) on {world} GPU(s), exchange {exchange}")
