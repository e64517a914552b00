"""bench.py on the CPU: the workload table matches BASELINE.json's configs, and the cpu_baseline legs (the genuine reference
codec on independent blocks, the OpenMP port, the serial reference) produce well-formed, round-trip-checked objects."""
import json
import os

import numpy as np
import pytest

import bench
from ndzip_amd.sharded import plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_table_covers_every_baseline_config():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        configs = json.load(f)["configs"]
    assert len(configs) == 5
    # per-GPU slabs times 8 GPUs reproduce configs[3] and configs[4]; config 5 is decompress-only
    _, dt4, slab4, _, mode4 = bench.CONFIGS["4"]
    assert (slab4[0] * 8,) + slab4[1:] == (2048, 1024, 1024) and dt4 == "float32" and mode4 == "both"
    _, dt5, slab5, _, mode5 = bench.CONFIGS["5"]
    assert (slab5[0] * 8,) + slab5[1:] == (1024, 1024, 1024) and dt5 == "float64" and mode5 == "decompress"
    assert bench.CONFIGS["1"][2] == (1 << 24,) and bench.CONFIGS["2"][2] == (512, 512, 512) and bench.CONFIGS["3"][2] == (8192, 8192)
    # the 16 GiB strong-scaling grid stays inside the format's uint32 counts and splits evenly over 1, 2, 4, 8 ranks
    g = bench.CONFIGS["16gib"][3]
    assert int(np.prod(g)) * 8 == 16 << 30 and int(np.prod(g)) < 2 ** 32
    for world in (1, 2, 4, 8):
        shards = plan_shards(g, world)
        assert len({s.num_hypercubes for s in shards}) == 1 and sum(s.num_hypercubes for s in shards) == int(np.prod(g)) // 4096
        assert shards[0].extent == (g[0] // world,) + g[1:]
    for world in (1, 2, 4, 8):  # cfg 4 / cfg 5 slabs at every scale the driver runs
        assert plan_shards((256 * world, 1024, 1024), world)[0].num_hypercubes == 65536
        assert plan_shards((128 * world, 1024, 1024), world)[0].num_hypercubes == 32768


@pytest.mark.parametrize("shape,threads", [((64, 64, 64), 8), ((32, 48, 16), 5), ((4096 * 5,), 3), ((256, 192), 7), ((16, 16, 16), 4)])
def test_blocks_partition_whole_hypercubes(shape, threads):
    blocks = bench._blocks(shape, threads)
    assert 1 <= len(blocks) <= threads
    side = {1: 4096, 2: 64, 3: 16}[len(shape)]
    seen = np.zeros(shape, dtype=np.int32)
    for b in blocks:
        assert all(s.start % side == 0 and s.stop % side == 0 for s in b)
        seen[b] += 1
    assert (seen == 1).all()


def _check_leg(leg, kind):
    assert set(leg) >= {"value", "unit", "cores", "kind", "sample"} and leg["unit"] == "GB/s" and leg["kind"] == kind
    assert leg["cores"] >= 1 and leg.get("roundtrip_ok", True)
    if leg["value"] is None:
        assert "reason" in leg  # a disturbed host: the field says so instead of a number
    else:
        assert leg["value"] > 0


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref is built from /root/reference (authoring container / shipped .so)")
def test_reference_legs_are_well_formed():
    grid = synth_numpy((64, 64, 64), np.float32, seed=1, noise_mask=0xFF)
    leg = bench.cpu_reference_blocks(grid, 4, budget_s=0.5)
    _check_leg(leg, "reference")
    assert leg["cores"] == 4 and len(leg["median_over_best"]) == 2
    s = bench.cpu_reference_serial(grid[:16], "test sample")
    _check_leg(s, "reference")
    assert s["cores"] == 1 and 0 < s["ratio"] < 1.1


def test_port_leg_is_well_formed():
    grid = synth_numpy((128, 128), np.float64, seed=2, noise_mask=0xFF)
    leg = bench.cpu_port_openmp(grid, 2, budget_s=0.5)
    _check_leg(leg, "port")
    s = bench.cpu_port_serial(grid, "test sample")
    _check_leg(s, "port")
    assert s["cores"] == 1 and 0 < s["ratio"] < 1.1 and s["roundtrip_ok"]


def test_configs0_has_a_cpu_line_without_the_compiled_reference(monkeypatch):
    """oracle/_ref does not reach every box (it is built from /root/reference): BASELINE configs[0] -- 1D float32 16 Mi, one CPU
    thread -- then still gets its line, timed on the C restatement and labelled as such"""
    monkeypatch.setattr(oracle, "have_ref", lambda: False)
    monkeypatch.setattr(bench, "cpu_port_openmp", lambda grid, cores, budget: {"value": 1.0, "unit": "GB/s", "cores": cores, "kind": "port", "sample": "stub"})
    real = bench.cpu_port_serial
    seen = {}

    def small(sample, what):  # (the 64 MiB array is what bench times on the box; the contract is checked on its first 64 Ki elements)
        seen["shape"], seen["dtype"] = sample.shape, sample.dtype
        return real(sample[: 1 << 16], what)

    monkeypatch.setattr(bench, "cpu_port_serial", small)
    legs = bench.cpu_legs(synth_numpy((16, 16, 16), np.float32, seed=1, noise_mask=0xFF), 0.2)
    assert seen == {"shape": (1 << 24,), "dtype": np.float32}
    assert legs["cpu_baseline"]["kind"] == "port" and legs["cpu_baseline"]["why_kind"].startswith("kind=port")
    leg = legs["cpu_port_serial_cfg1"]
    _check_leg(leg, "port")
    assert leg["cores"] == 1 and "configs[0]" in leg["sample"] and "cpu_reference_serial_cfg1" not in legs


class _HostAccelerator:
    """Stand-in for bench.Accelerator: host tensors, wall-clock "events" (the kernels run on the functional model)."""

    def __init__(self, index):
        import torch

        self.device = torch.device("cpu")

    def synchronize(self):
        pass

    def event(self):
        import time

        class _Ev:
            t = 0.0

            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return max(1e-6, (other.t - self.t) * 1e3)

        return _Ev()


@pytest.mark.parametrize("argv,mode", [(["--shape", "32,32,32"], "both"), (["--shape", "128,128", "--dtype", "float64", "--decompress-only"], "decompress"),
                                       (["--shape", "8192", "--compress-only", "--data", "random"], "compress"),
                                       (["--shape", "128,128", "--dtype", "float64", "--decompress-only", "--f64-work-items", "256"], "decompress")])
def test_main_produces_the_contract_line_on_the_model(monkeypatch, capsys, argv, mode):
    """bench.main() end to end -- workload set-up, the timed loop, verification, the JSON line with roofline and cpu_baseline --
    with the kernels on the wave64 functional model (tests/wavesim) and host tensors: every key the driver and the judge read
    is there and consistent.  (Numbers from this run mean nothing; the benchmark proper needs the GPU.)"""
    from tests.wavesim import sim

    monkeypatch.setattr(bench, "Accelerator", _HostAccelerator)
    with sim.active():
        bench.main(argv + ["--steps", "2", "--warmup", "1", "--cpu-budget", "0.3"])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["ranks"] == {"world_size": 1, "backend": None, "device_ids": [0], "launched_by": "python"} and d["config"]["host"].startswith("Python")
    # (the model is ~10^5 times slower than the GPU: the rounded GB/s figure may well be 0.0 here)
    assert d["unit"] == "GB/s" and d["value"] >= 0 and d["ms_per_step"] > 0 and d["scaling"] == "weak" and d["roundtrip_bit_exact"] is True
    assert set(d["config"]) >= {"workload", "hypercubes", "compression_ratio", "step"} and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert ("decompress_kernel" in r["kernel"]) == (mode == "decompress")
    if mode == "decompress":  # (64-bit profiles decode with 128 work-items per hypercube unless the A/B switch says 256)
        assert r["kernel"] == ("decompress_kernel_wide<2>" if "--f64-work-items" in argv else "decompress_kernel<double,2>")
    assert ("decompress" in r) == (mode == "both")
    assert set(d["per_gpu"]) == {"both": {"compress_GBps", "compress_frac_of_hbm_peak", "decompress_GBps", "decompress_frac_of_hbm_peak"},
                                 "compress": {"compress_GBps", "compress_frac_of_hbm_peak"},
                                 "decompress": {"decompress_GBps", "decompress_frac_of_hbm_peak"}}[mode]
    assert "traffic" in r and r["traffic"] is None  # (no counters were ever taken on a 32^3 model run: null, not a stale number)
    cb = d["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample", "why_kind"} and cb["kind"] in ("reference", "port")
    assert cb["why_kind"].startswith("kind=" + cb["kind"])
    if oracle.have_ref():
        assert cb["kind"] == "reference" and "cpu_reference_serial_cfg1" in d and d["cpu_reference_serial_cfg1"]["cores"] == 1


def test_native_exchange_single_rank_on_the_model(monkeypatch, capsys):
    """bench.py --native-exchange at N = 1: the step goes through libndzip_hip_rccl's driver (here: sharded.cc on the model), the line
    is the same contract line and says which host produced it."""
    import ctypes

    from ndzip_amd import sharded_native
    from tests.test_sharded_native_cpu import _gloo_table
    from tests.wavesim import build as simbuild
    from tests.wavesim import sim

    monkeypatch.setattr(bench, "Accelerator", _HostAccelerator)
    monkeypatch.setattr(sharded_native, "_lib", sharded_native._bind(ctypes.CDLL(simbuild.build_sharded()), rccl=False))
    real = sharded_native.NativeShardedCodec
    monkeypatch.setattr(sharded_native, "NativeShardedCodec", lambda *a, **k: real(*a, collectives=_gloo_table(1)[0], **k))
    with sim.active():
        bench.main(["--shape", "130,200", "--dtype", "float64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--native-exchange"])
        with pytest.raises(SystemExit, match="--native-exchange has no"):
            bench.main(["--shape", "32,32,32", "--native-exchange", "--workgroups-per-cu", "2"])  # (an A/B handle of the Python driver)
    d = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert d["roundtrip_bit_exact"] is True and d["n_gpus"] == 1 and d["config"]["host"].startswith("C++ (libndzip_hip_rccl.so")
    # the same workload through the Python driver: same stream length, hence the same ratio
    with sim.active():
        bench.main(["--shape", "130,200", "--dtype", "float64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    d2 = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert d2["config"]["compression_ratio"] == d["config"]["compression_ratio"] and d2["config"]["host"].startswith("Python")


def test_traffic_is_reported_only_for_counters_taken_on_these_kernels(monkeypatch, capsys, tmp_path):
    """profiles/traffic.json entries carry the fingerprint of the device-code sources they were measured on; an entry of other
    kernels (the committed round-1 one, say) leaves roofline.traffic null and says why."""
    import shutil

    from ndzip_amd.build import kernels_fingerprint
    from tests.wavesim import sim

    root = tmp_path / "root"
    (root / "profiles").mkdir(parents=True)
    monkeypatch.setattr(bench, "Accelerator", _HostAccelerator)
    monkeypatch.setattr(bench, "ROOT", str(root))
    shutil.copytree(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "oracle"), root / "oracle", symlinks=True)
    for kernels, want in (("somebody-else", None), (kernels_fingerprint(), 4242)):
        with open(root / "profiles" / "traffic.json", "w") as f:
            json.dump({"float32-32x32x32": {"kernels": kernels, "compress_hbm_bytes_per_launch": 4242, "source": "test"}}, f)
        with sim.active():
            bench.main(["--shape", "32,32,32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
        d = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
        assert d["roofline"]["traffic"] == want
        assert ("none:" in d["roofline"].get("traffic_source", "")) == (want is None)


def _bench_rank(rank, world, port, out_dir):
    import contextlib
    import io

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NDZIP_BENCH_SHARE_GPU="1")  # -> gloo group (RCCL refuses two ranks on one device; here there is none at all)
    from tests.wavesim import sim

    bench.Accelerator = _HostAccelerator
    buf = io.StringIO()
    with sim.active(), contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--shape", "32,32,32", "--steps", "2", "--warmup", "1"])
    with open(os.path.join(out_dir, f"rank{rank}.out"), "w") as f:
        f.write(buf.getvalue())


def test_main_with_two_ranks_over_gloo_on_the_model(tmp_path):
    """The N > 1 path of bench.py as the driver launches it (one process per rank, RANK / WORLD_SIZE from the environment):
    the sharded codec with its two collectives, max-over-ranks timing, sums over ranks -- rank 0 alone prints the line."""
    import socket

    import torch.multiprocessing as mp

    from tests.wavesim import build as simbuild

    simbuild.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bench_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out0, out1 = (tmp_path / "rank0.out").read_text(), (tmp_path / "rank1.out").read_text()
    assert not [l for l in out1.splitlines() if l.startswith("{")], "only rank 0 prints"
    d = json.loads([l for l in out0.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roundtrip_bit_exact"] is True and "cpu_baseline" not in d
    assert d["config"]["hypercubes"] == 2 * 8 and "64x32x32" in d["config"]["workload"] and "2 z-slab(s) of 32x32x32" in d["config"]["workload"]


@pytest.mark.parametrize("extra", [[], ["--native-exchange"], ["--native-exchange", "--overlap-exchange"]], ids=["python-driver", "cpp-host", "cpp-host-overlap"])
def test_gpus_2_typed_as_is_starts_its_own_ranks(tmp_path, extra):
    """`python bench.py --gpus 2 ...` with NO RANK / WORLD_SIZE in the environment (what the driver's single-command form and a
    user at a shell type): bench.py starts the two ranks itself under torch.distributed.run, stdout carries exactly rank 0's
    line, the exit status is the ranks'.  Here the ranks run tests/bench_on_model.py (kernels on the functional model, gloo)."""
    import subprocess
    import sys

    from tests.wavesim import build as simbuild

    simbuild.build()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NDZIP_BENCH_SHARE_GPU")}
    env["NDZIP_BENCH_ENTRY"] = os.path.join(ROOT, "tests", "bench_on_model.py")
    simbuild.build_sharded()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "32,32,32", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["roundtrip_bit_exact"] is True and "cpu_baseline" not in d
    assert d["ranks"] == {"world_size": 2, "backend": "gloo", "device_ids": [0, 0], "launched_by": "bench.py (torch.distributed.run child)"}
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr and "127.0.0.1" in r.stderr
    assert d["config"]["host"].startswith("C++ (libndzip_hip_rccl.so" if extra else "Python (ndzip_amd.sharded.ShardedCodec")
    if extra:
        return
    # a failing rank is the launcher's failure too, and a silent rank 0 (exit 0, no line) is not a success either
    for body, what in (("import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)", "rank 1 fails"), ("pass", "nobody prints")):
        entry = tmp_path / "entry.py"
        entry.write_text(body + "\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "32,32,32"],
                           capture_output=True, text=True, env=dict(env, NDZIP_BENCH_ENTRY=str(entry)), cwd=str(tmp_path), timeout=600)
        assert r.returncode != 0 and not r.stdout.strip(), what
