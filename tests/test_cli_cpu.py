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
    for opt in ("--decompress", "--array-size", "--data-type", "--target_str", "--threads", "--input", "--output", "--no-mmap", "--mmap"):
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


@pytest.fixture(scope="module")
def model_cli(tmp_path_factory):
    """The file tool linked against the wave64 functional model of the library (tests/wavesim, test infrastructure): the whole
    tool -- option parsing, mapped and buffered I/O, the pipelined chunk loop, the C ABI and the kernels' logic -- on the CPU."""
    from tests.wavesim import build as simbuild

    lib = simbuild.build()
    exe = str(tmp_path_factory.mktemp("cli") / "ndzip-hip-model")
    src = os.path.join(os.path.dirname(build.__file__), "cli", "ndzip_hip_cli.cc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, src, lib, "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("io", ["mmap", "no-mmap", "default", "pipes"])
@pytest.mark.parametrize("dtype,shape,chunks", [(np.float32, (32, 32, 48), 4), (np.float64, (70, 130), 3), (np.float32, (4096 * 2 + 9,), 5)])
def test_file_roundtrip_matches_the_reference_format(model_cli, tmp_path, io, dtype, shape, chunks):
    """compress.cc:17-86: the output is the plain concatenation of one stream per array; files are interchangeable with the
    reference tool's (= the oracle's streams); mapped and buffered I/O (src/io/io.cc) give identical files."""
    from ndzip_amd.synth import synth_numpy

    arrays = [synth_numpy(shape, dtype, seed=10 + i, noise_mask=0xFF if i % 2 else 0xFFFFF) for i in range(chunks)]
    arrays[1][...] = 0  # a highly compressible array next to poorly compressible ones: in-flight output windows must move up
    raw = np.concatenate([a.reshape(-1) for a in arrays])
    want = np.concatenate([oracle.compress(a) for a in arrays])
    src, ndz, back = tmp_path / "in.bin", tmp_path / "out.ndz", tmp_path / "back.bin"
    raw.tofile(src)
    size = [str(x) for x in shape]
    t = ["-t", "double"] if dtype == np.float64 else []
    if io == "pipes":
        r = run(model_cli, "-n", *size, *t, stdin=raw.tobytes())
        assert r.returncode == 0, r.stderr
        got = np.frombuffer(r.stdout, dtype=want.dtype)
        r2 = run(model_cli, "-d", "-n", *size, *t, stdin=r.stdout)
        assert r2.returncode == 0, r2.stderr
        out = np.frombuffer(r2.stdout, dtype=dtype)
    else:
        extra = ["--no-mmap"] if io == "no-mmap" else ["--mmap"] if io == "mmap" else []
        r = run(model_cli, "-n", *size, *t, "-i", str(src), "-o", str(ndz), *extra)
        assert r.returncode == 0, r.stderr
        assert b"ratio" in r.stderr and f"({chunks} chunks".encode() in r.stderr
        got = np.fromfile(ndz, dtype=want.dtype)
        r2 = run(model_cli, "-d", "-n", *size, *t, "-i", str(ndz), "-o", str(back), *extra)
        assert r2.returncode == 0, r2.stderr
        out = np.fromfile(back, dtype=dtype)
    assert len(got) == len(want) and np.array_equal(got, want)
    assert np.array_equal(out.view(want.dtype), raw.view(want.dtype))


def test_file_tool_rejects_partial_and_corrupt_input(model_cli, tmp_path):
    src = tmp_path / "in.bin"
    np.zeros(4096 + 5, dtype=np.float32).tofile(src)
    r = run(model_cli, "-n", "4096", "-i", str(src), "-o", str(tmp_path / "o.ndz"))
    assert r.returncode != 0 and b"not a multiple of the chunk size" in r.stderr
    good = oracle.compress(np.ones(4096 * 2, dtype=np.float32))
    bad = good.copy()
    bad[0] = 0xFFFFFF00  # first header entry
    # (a truncated file: the buffered reader notices while refilling, the mapped one when the header is checked against the map)
    for blob, msg, io in ((bad, b"corrupt stream header", []), (bad, b"corrupt stream header", ["--mmap"]),
                          (good[:-3], b"truncated stream in input", []), (good[:-3], b"longer than the given words", ["--mmap"])):
        f = tmp_path / "bad.ndz"
        blob.tofile(f)
        r = run(model_cli, "-d", "-n", "8192", "-i", str(f), "-o", str(tmp_path / "b.bin"), *io)
        assert r.returncode != 0 and msg in r.stderr, r.stderr


def test_benchmark_tool_on_the_model(tmp_path):
    """ndzip-hip-benchmark (the `ndzip-hip` rows of the reference's result CSV, benchmark.cc:1332-1337,1487-1489) linked against the
    functional model: dataset CSV in, header and one row per dataset out, sizes against the oracle, round trip verified by the tool."""
    from ndzip_amd.synth import synth_numpy
    from tests.wavesim import build as simbuild

    lib = simbuild.build()
    exe = str(tmp_path / "ndzip-hip-benchmark-model")
    src = os.path.join(os.path.dirname(build.__file__), "cli", "ndzip_hip_benchmark.cc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, src, lib, "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sets = [("a.f32", np.float32, (70, 130)), ("b.f64", np.float64, (16, 40, 17)), ("c.f32", np.float32, (4096 * 3 + 5,))]
    lines, want = [], {}
    for name, dtype, shape in sets:
        data = synth_numpy(shape, dtype, seed=11, noise_mask=0xFF)
        data.tofile(tmp_path / name)
        lines.append(f"{name};{'float' if dtype == np.float32 else 'double'};{' '.join(str(x) for x in shape)}")
        want[name] = (data.nbytes, oracle.compress(data).nbytes, len(shape))
    (tmp_path / "sets.csv").write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, "-r", "2", "-t", "0", str(tmp_path / "sets.csv")], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    rows = r.stdout.decode().strip().split("\n")
    assert rows[0].split(";") == ["dataset", "data type", "dimensions", "algorithm", "tunable", "number of threads",
                                  "compression times (microseconds)", "decompression times (microseconds)", "uncompressed bytes",
                                  "compressed bytes"]
    assert len(rows) == 1 + len(sets)
    for row in rows[1:]:
        c = row.split(";")
        raw, comp, dims = want[c[0]]
        assert c[1] in ("float", "double") and int(c[2]) == dims and c[3] == "ndzip-hip" and c[4] == "1" and c[5] == "1"
        assert len(c[6].split(",")) >= 2 and len(c[7].split(",")) >= 2
        assert int(c[8]) == raw and int(c[9]) == comp


@pytest.mark.skipif(not os.path.exists("/root/reference/src/benchmark/plot_benchmark.py"), reason="the reference tree only exists in the authoring container")
def test_reference_plot_script_reads_our_benchmark_rows(tmp_path):
    """the rows ndzip-hip-benchmark prints are fed to the reference's own plot_benchmark.py UNCHANGED (run from where
    it lies, headless); it must parse them and tabulate an `ndzip-hip` line for both value types."""
    pytest.importorskip("matplotlib")
    pytest.importorskip("tabulate")
    from ndzip_amd.synth import synth_numpy
    from tests.wavesim import build as simbuild

    lib = simbuild.build()
    exe = str(tmp_path / "ndzip-hip-benchmark-model")
    src = os.path.join(os.path.dirname(build.__file__), "cli", "ndzip_hip_benchmark.cc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, src, lib, "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = []
    for name, dtype, shape in [("a.f32", np.float32, (64, 128)), ("b.f32", np.float32, (32, 32, 32)), ("c.f64", np.float64, (64, 64)), ("d.f64", np.float64, (8192,))]:
        synth_numpy(shape, dtype, seed=3, noise_mask=0xFF).tofile(tmp_path / name)
        lines.append(f"{name};{'float' if dtype == np.float32 else 'double'};{' '.join(str(x) for x in shape)}")
    (tmp_path / "sets.csv").write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, "-r", "3", "-t", "0", str(tmp_path / "sets.csv")], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    (tmp_path / "rows.csv").write_bytes(r.stdout)
    p = subprocess.run([os.sys.executable, "/root/reference/src/benchmark/plot_benchmark.py", str(tmp_path / "rows.csv")], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, MPLBACKEND="Agg"), cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("ndzip-hip 1") >= 2 and "(float)" in p.stdout and "(double)" in p.stdout and "MB/s" in p.stdout
