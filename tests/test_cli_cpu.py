"""CPU suite for the file-level tool ndzip_amd/ndzip-hip (the reference's src/compress for this back-end): it builds, parses the
reference's options (compress.cc:135-211), rejects what it cannot do, and -- without a GPU -- fails loudly.  Plus the host-only
stream splitter ndzip_hip_stream_words against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import ndzip_amd
from ndzip_amd import build
from oracle import oracle


@pytest.fixture(scope="module")
def cli():
    ndzip_amd.hip.lib()
    return build.build_cli()


def run(cli, *args, stdin=b""):
    return subprocess.run([cli, *args], input=stdin, capture_output=True, timeout=120)


def test_help_lists_the_reference_options(cli):
    r = run(cli, "--help")
    text = (r.stdout + r.stderr).decode()
    for opt in ("--decompress", "--array-size", "--data-type", "--target_str", "--threads", "--input", "--output", "--no-mmap"):
        assert opt in text
    assert "Compress or decompress binary float dump" in text


def test_argument_errors(cli):
    assert b"'--array-size' is required" in run(cli).stderr
    assert b"Unimplemented target cpu" in run(cli, "-n", "64", "-e", "cpu").stderr          # compress.cc:188
    assert b"Invalid data type half" in run(cli, "-n", "64", "-t", "half").stderr            # compress.cc:203
    assert b"Expected between 1 and 3 dimensions, got 4" in run(cli, "-n", "2", "3", "4", "5").stderr  # compress.cc:191-193
    assert b"unrecognised option" in run(cli, "-n", "64", "--frobnicate").stderr
    for r in (run(cli), run(cli, "-n", "64", "-e", "cpu")):
        assert r.returncode != 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_a_device(cli, tmp_path):
    f = tmp_path / "in.bin"
    np.zeros(4096, dtype=np.float32).tofile(f)
    r = run(cli, "-n", "4096", "-i", str(f), "-o", str(tmp_path / "out.ndz"))
    assert r.returncode != 0 and b"no CPU fallback" in r.stderr


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_stream_words_splits_concatenated_streams(dtype):
    """compress.cc:62-86 walks a file of concatenated streams; the splitter must find each length from the header alone."""
    rng = np.random.default_rng(5)
    shapes = [(4096 * 3 + 7,), (70, 200), (17, 35, 33), (5,), (64, 64)]
    for shape in shapes:
        chunks = [rng.random(shape).astype(dtype), np.zeros(shape, dtype=dtype), (rng.random(shape) * 1e-3).astype(dtype)]
        streams = [oracle.compress(c) for c in chunks]
        blob = np.concatenate(streams)
        pos = 0
        for s in streams:
            assert ndzip_amd.stream_words(dtype, shape, blob[pos:]) == len(s)
            pos += len(s)
        assert pos == len(blob)
        # a header that points past the available words is rejected, not trusted
        if oracle.num_hypercubes(shape):
            with pytest.raises(ndzip_amd.NdzipHipError):
                ndzip_amd.stream_words(dtype, shape, streams[0][: len(streams[0]) - 1])


def test_benchmark_tool_parses_the_reference_dataset_csv(tmp_path):
    """src/benchmark/benchmark.cc:102-125: `name;float|double;n0 [n1 [n2]]`; malformed lines are fatal; header line is printed."""
    ndzip_amd.hip.lib()
    build.build_cli()
    tool = build.BENCHMARK_OUT
    bad = tmp_path / "bad.csv"
    bad.write_text("a.bin;int;4096\n")
    r = subprocess.run([tool, str(bad)], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"Invalid line" in r.stderr
    r = subprocess.run([tool], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"csv-file" in r.stderr
    r = subprocess.run([tool, "-a", "zfp", str(bad)], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"unknown algorithm" in r.stderr
    empty = tmp_path / "empty.csv"
    empty.write_text("")
    r = subprocess.run([tool, str(empty)], capture_output=True, timeout=120)
    assert r.returncode == 0
    assert r.stdout.decode().strip() == ("dataset;data type;dimensions;algorithm;tunable;number of threads;"
                                         "compression times (microseconds);decompression times (microseconds);"
                                         "uncompressed bytes;compressed bytes")
