"""GPU suite: the file-level tool and the pipelined host-pointer offloader against the oracle.

A file written by the reference tool is the plain concatenation of one stream per `-n` array (compress.cc:34-45); the oracle's
streams are bit-identical to the compiled reference's, so `concat(oracle.compress(chunk))` IS the reference tool's output."""
import subprocess

import numpy as np
import pytest

import ndzip_amd
from ndzip_amd import build
from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import same_bits

pytestmark = pytest.mark.gpu

CASES = [(np.float32, (4096 * 2 + 3,), 5), (np.float32, (70, 130), 4), (np.float32, (33, 20, 50), 7), (np.float64, (64, 200), 3),
         (np.float64, (16, 40, 17), 5), (np.float32, (7,), 3)]


def _chunks(dtype, shape, n):
    return [synth_numpy(shape, dtype, seed=100 + i, noise_mask=0xFF if i % 2 else 0xFFFF) for i in range(n)]


@pytest.fixture(scope="module")
def cli(hiplib):
    return build.build_cli()


@pytest.mark.hardware_only  # (the binaries link the real library; tests/test_cli_cpu.py runs them linked against the model)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{np.dtype(c[0]).name}-{len(c[1])}d-x{c[2]}")
@pytest.mark.parametrize("slots,mmap", [(3, False), (1, False), (1, True), (3, True)], ids=["3slots-stdio", "1slot-stdio", "1slot-mmap", "3slots-mmap"])
def test_cli_files_are_the_reference_tools_files(cli, cuda_device, tmp_path, case, slots, mmap):
    dtype, shape, n = case
    io = ["--mmap"] if mmap else ["--no-mmap"]  # src/io/io.cc: mapped files (here opt-in) or stdio (and for pipes)
    chunks = _chunks(dtype, shape, n)
    raw = tmp_path / "in.bin"
    np.concatenate([c.reshape(-1) for c in chunks]).tofile(raw)
    want = np.concatenate([oracle.compress(c) for c in chunks])
    size = [str(x) for x in shape]
    t = "float" if dtype == np.float32 else "double"
    ndz, back = tmp_path / "out.ndz", tmp_path / "back.bin"
    r = subprocess.run([cli, "-n", *size, "-t", t, "-e", "hip", "-i", str(raw), "-o", str(ndz), "--slots", str(slots), *io], capture_output=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert b"ratio = " in r.stderr and (n == 1 or f"({n} chunks".encode() in r.stderr)
    got = np.fromfile(ndz, dtype=want.dtype)
    assert got.size == want.size and np.array_equal(got, want)
    # ... and a file from the reference tool (== `want`) decompresses to the input, through stdin/stdout
    r = subprocess.run([cli, "-d", "-n", *size, "-t", t, "--slots", str(slots)], input=want.tobytes(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert same_bits(np.frombuffer(r.stdout, dtype=dtype), np.concatenate([c.reshape(-1) for c in chunks]))
    r = subprocess.run([cli, "-d", "-n", *size, "-t", t, "-i", str(ndz), "-o", str(back), *io], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert same_bits(np.fromfile(back, dtype=dtype), np.concatenate([c.reshape(-1) for c in chunks]))


@pytest.mark.hardware_only
def test_cli_rejects_partial_and_truncated_input(cli, cuda_device, tmp_path):
    raw = tmp_path / "in.bin"
    np.arange(4096 + 10, dtype=np.float32).tofile(raw)
    r = subprocess.run([cli, "-n", "4096", "-i", str(raw), "-o", str(tmp_path / "o")], capture_output=True, timeout=300)
    assert r.returncode != 0 and b"not a multiple of the chunk size" in r.stderr  # io.cc read_exact
    stream = oracle.compress(np.arange(4096 * 2, dtype=np.float32))
    r = subprocess.run([cli, "-d", "-n", "8192"], input=stream.tobytes()[:-8], capture_output=True, timeout=300)
    assert r.returncode != 0 and b"truncated" in r.stderr


@pytest.mark.parametrize("case", CASES[:5], ids=lambda c: f"{np.dtype(c[0]).name}-{len(c[1])}d")
def test_pipelined_offloader_matches_oracle(hiplib, cuda_device, case):
    dtype, shape, _ = case
    n_jobs, slots = 7, 3
    chunks = _chunks(dtype, shape, n_jobs)
    off = ndzip_amd.HipPipelinedOffloader(dtype, shape, slots=slots)
    wdt = ndzip_amd.word_dtype(dtype)
    bound = ndzip_amd.compressed_length_bound(dtype, shape)
    ins = [ndzip_amd.PinnedBuffer(chunks[0].nbytes, dtype) for _ in range(slots)]
    outs = [ndzip_amd.PinnedBuffer(bound * np.dtype(wdt).itemsize, wdt) for _ in range(slots)]
    streams = []

    def retire(j):
        words, ns = off.wait(j % slots)
        assert ns > 0
        streams.append(outs[j % slots].array[:words].copy())

    for j, c in enumerate(chunks):
        if j >= slots:
            retire(j - slots)
        ins[j % slots].array[:] = c.reshape(-1)
        off.submit_compress(j % slots, ins[j % slots].array.reshape(shape), outs[j % slots].array)
    for j in range(max(0, n_jobs - slots), n_jobs):
        retire(j)
    for c, s in zip(chunks, streams):
        assert np.array_equal(s, oracle.compress(c))
    # decompress through the same handle, pageable buffers this time
    with pytest.raises(ndzip_amd.NdzipHipError, match="busy|idle"):
        off.wait(0)
    results = [np.zeros(shape, dtype=dtype) for _ in range(n_jobs)]
    for j, s in enumerate(streams):
        if j >= slots:
            assert off.wait((j - slots) % slots)[0] == len(streams[j - slots])
        off.submit_decompress(j % slots, s, results[j])
    for j in range(max(0, n_jobs - slots), n_jobs):
        assert off.wait(j % slots)[0] == len(streams[j])
    for c, r in zip(chunks, results):
        assert same_bits(r, c)
    with pytest.raises(ndzip_amd.NdzipHipError):
        off.submit_compress(0, np.zeros(tuple(2 * x for x in shape), dtype=dtype), outs[0].array)  # larger than created for
    off.close()
    for b in ins + outs:
        b.close()


@pytest.mark.hardware_only
def test_benchmark_tool_writes_the_reference_result_csv(hiplib, cuda_device, tmp_path):
    """Column set and row format of src/benchmark/benchmark.cc:1332-1337,1487-1489, sizes against the oracle."""
    sets = [("a.f32", np.float32, (70, 130)), ("b.f64", np.float64, (16, 40, 17)), ("c.f32", np.float32, (4096 * 3 + 5,))]
    lines = []
    want = {}
    for name, dtype, shape in sets:
        data = synth_numpy(shape, dtype, seed=11, noise_mask=0xFF)
        data.tofile(tmp_path / name)
        lines.append(f"{name};{'float' if dtype == np.float32 else 'double'};{' '.join(str(x) for x in shape)}")
        want[name] = (data.nbytes, oracle.compress(data).nbytes, len(shape))
    (tmp_path / "sets.csv").write_text("\n".join(lines) + "\n")
    build.build_cli()
    r = subprocess.run([build.BENCHMARK_OUT, "-r", "3", "-t", "1", str(tmp_path / "sets.csv")], capture_output=True, timeout=600)
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
        ct, dt = [int(x) for x in c[6].split(",")], [int(x) for x in c[7].split(",")]
        assert len(ct) >= 3 and len(dt) >= 3 and all(x >= 0 for x in ct + dt)
        assert int(c[8]) == raw and int(c[9]) == comp
