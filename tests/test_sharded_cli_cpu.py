"""CPU suite for ndzip_amd/ndzip-hip-sharded (ndzip_amd/cli/ndzip_hip_sharded_cli.cc): ONE array as ONE ndzip stream over N ranks --
the file-level tool of the multi-GPU path.  It builds against the real libraries, parses its options, fails loudly without a GPU; and,
linked against the kernels' functional model (ranks = threads, the library's in-process exchange), it writes the oracle's stream
byte for byte for 1 / 2 / 3 / 4 / 8 ranks, reads reference streams back, and its files are interchangeable with the single-GPU tool's.
On hardware: tests/test_hip_sharded_native.py."""
import os
import subprocess

import numpy as np
import pytest

from ndzip_amd import build
from ndzip_amd.synth import synth_numpy
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(np.float32, (50, 37, 41), 3), (np.float64, (64 * 5 + 5, 130), 4), (np.float32, (6 * 4096 + 5,), 3), (np.float64, (32, 16, 48), 1),
         (np.float32, (10, 70), 2), (np.float32, (128, 32, 32), 8), (np.float64, (4096 * 2 + 3,), 2)]


def run(exe, *args, env=None):
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=env)


@pytest.fixture(scope="module")
def real_cli():
    build.build()
    return build.SHARDED_CLI_OUT


def test_options_and_loud_failure_without_a_gpu(real_cli, tmp_path):
    text = run(real_cli, "--help")
    assert text.returncode == 0 and all(o in text.stderr for o in ("--decompress", "--array-size", "--data-type", "--ranks", "--devices", "--exchange", "--repeat"))
    assert "'--array-size' is required" in run(real_cli).stderr
    assert "Invalid data type half" in run(real_cli, "-n", 64, "-t", "half", "-i", "a", "-o", "b").stderr            # compress.cc:203
    assert "Expected between 1 and 3 dimensions, got 4" in run(real_cli, "-n", 2, 3, 4, 5, "-i", "a", "-o", "b").stderr  # compress.cc:191-193
    assert "Invalid exchange mpi" in run(real_cli, "-n", 64, "--exchange", "mpi", "-i", "a", "-o", "b").stderr
    assert "unrecognised option" in run(real_cli, "-n", 64, "--frobnicate").stderr
    if not os.path.exists("/dev/kfd"):
        f = tmp_path / "in.bin"
        np.zeros(4096, dtype=np.float32).tofile(f)
        r = run(real_cli, "-n", 4096, "-i", f, "-o", tmp_path / "out.ndz")
        assert r.returncode != 0 and "no CPU fallback" in r.stderr
    needed = subprocess.run(["readelf", "-d", real_cli], capture_output=True, text=True).stdout
    assert "libndzip_hip_rccl.so" in needed and "amdhip64" not in needed  # (plain C++ over the C ABI: no HIP runtime of its own)


@pytest.fixture(scope="module")
def model_cli(tmp_path_factory):
    """The tool linked against the functional model of the library (test infrastructure): no RCCL there, so `--exchange local`."""
    from tests.wavesim import build as simbuild

    lib = simbuild.build_sharded(variant="")
    here = os.path.dirname(lib)
    exe = str(tmp_path_factory.mktemp("cli") / "ndzip-hip-sharded-model")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Werror", "-o", exe, build.SHARDED_CLI_SRC, "-L" + here, "-l:" + os.path.basename(lib),
                        "-l:libndzip_hip_wavesim.so", "-Wl,-rpath," + here], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


MODEL_ENV = dict(os.environ, WAVESIM_CUS="2", WAVESIM_BLOCKS_PER_CU="2")
MODEL_ENV.pop("WAVESIM_VARIANT", None)


@pytest.mark.parametrize("dtype,extent,ranks", CASES)
def test_one_array_one_stream_n_ranks_on_the_model(model_cli, tmp_path, dtype, extent, ranks):
    data = synth_numpy(extent, dtype, seed=11, noise_mask=0xFF)
    want = oracle.compress(data)
    src, ndz, ref, back = tmp_path / "in.bin", tmp_path / "out.ndz", tmp_path / "ref.ndz", tmp_path / "back.bin"
    data.tofile(src)
    want.tofile(ref)
    t = ["-t", "float" if np.dtype(dtype).itemsize == 4 else "double"]
    r = run(model_cli, "-n", *extent, *t, "-i", src, "-o", ndz, "--ranks", ranks, "--repeat", 2, env=MODEL_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(ndz, dtype=want.dtype)
    assert len(got) == len(want) and np.array_equal(got, want), "the ranks' pieces do not add up to the reference stream"
    assert f"{ranks} rank(s) on 1 GPU(s), exchange {'none' if ranks == 1 else 'local'}" in r.stderr and f"compressed = {want.nbytes} bytes" in r.stderr
    # the way back, from the REFERENCE's stream, with another number of ranks than it was (not) written with
    for n in {ranks, 1, 3}:
        r = run(model_cli, "-d", "-n", *extent, *t, "-i", ref, "-o", back, "--ranks", n, "-q", env=MODEL_ENV)
        assert r.returncode == 0 and not r.stderr, r.stderr[-2000:]
        assert np.array_equal(np.fromfile(back, dtype=want.dtype), data.reshape(-1).view(want.dtype))


def test_rejects_what_is_not_one_array_or_one_stream(model_cli, tmp_path):
    data = synth_numpy((70, 130), np.float32, seed=2, noise_mask=0xFF)
    src, ndz = tmp_path / "in.bin", tmp_path / "out.ndz"
    data.tofile(src)
    r = run(model_cli, "-n", 70, 131, "-i", src, "-o", ndz, "--ranks", 2, env=MODEL_ENV)
    assert r.returncode != 0 and "one array per file" in r.stderr
    r = run(model_cli, "-n", 70, 130, "-i", src, "-o", ndz, "--ranks", 2, "--exchange", "rccl", env=MODEL_ENV)
    assert r.returncode != 0 and "no RCCL transport" in r.stderr            # (the model build of the library: sharded.cc only)
    r = run(model_cli, "-n", 70, 130, "-i", src, "-o", ndz, "--devices", 2, env=MODEL_ENV)
    assert r.returncode != 0 and "1 GPU(s) are visible" in r.stderr
    want = oracle.compress(data)
    bad = want.copy()
    bad[1] = bad[0]  # second header entry does not follow the first by a hypercube's length
    for blob in (want[:-7], bad):
        blob.tofile(ndz)
        r = run(model_cli, "-d", "-n", 70, 130, "-i", ndz, "-o", tmp_path / "b.bin", "--ranks", 2, env=MODEL_ENV)
        assert r.returncode != 0 and "load" in r.stderr, r.stderr[-500:]
    r = run(model_cli, "-n", 65536, 65536, "-i", src, "-o", ndz, env=MODEL_ENV)   # 2^32 elements: not an ndzip array
    assert r.returncode != 0


def test_files_are_interchangeable_with_the_single_gpu_tool(model_cli, tmp_path):
    """ndzip-hip (the reference `compress` tool's options and file format, one device) reads what ndzip-hip-sharded wrote and the
    other way round -- both on the model."""
    from tests.wavesim import build as simbuild

    lib = simbuild.build()
    single = str(tmp_path / "ndzip-hip-model")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", single, os.path.join(ROOT, "ndzip_amd", "cli", "ndzip_hip_cli.cc"), "-L" + os.path.dirname(lib),
                        "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    extent, dtype = (96, 40, 33), np.float64
    data = synth_numpy(extent, dtype, seed=4, noise_mask=0xFF)
    src, a, b, back = tmp_path / "in.bin", tmp_path / "sharded.ndz", tmp_path / "single.ndz", tmp_path / "back.bin"
    data.tofile(src)
    size = [str(x) for x in extent]
    assert run(model_cli, "-n", *size, "-t", "double", "-i", src, "-o", a, "--ranks", 3, "-q", env=MODEL_ENV).returncode == 0
    assert run(single, "-n", *size, "-t", "double", "-i", src, "-o", b, "-q", env=MODEL_ENV).returncode == 0
    assert a.read_bytes() == b.read_bytes()
    assert run(single, "-d", "-n", *size, "-t", "double", "-i", a, "-o", back, "-q", env=MODEL_ENV).returncode == 0
    assert back.read_bytes() == src.read_bytes()
    assert run(model_cli, "-d", "-n", *size, "-t", "double", "-i", b, "-o", back, "--ranks", 2, "-q", env=MODEL_ENV).returncode == 0
    assert back.read_bytes() == src.read_bytes()
