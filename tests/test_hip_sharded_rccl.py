"""N-rank parity of the sharded path over RCCL: one process per visible GPU (torch.distributed.run, backend "nccl"), every rank
runs ndzip_amd.sharded.ShardedCodec on its slab -- compress x3 on one handle, decompress -- and the stream assembled from the
ranks' resident bodies and the all-gathered header must be byte-identical to the oracle's stream of the WHOLE array
(the reference has no multi-device path, README.md:13-14 -- its single-device stream is the contract).

Needs >= 2 GPUs: skipped (not failed) on the 1-GPU test box.  The same rank script is rehearsed over gloo on the kernels'
functional model in tests/test_sharded_cpu.py, so that the first multi-GPU node runs a harness that is known to work."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ndzip_amd.sharded import assemble_stream, plan_shards
from oracle import oracle
from tests.mp.sharded_rank_main import case_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANK_MAIN = os.path.join(ROOT, "tests", "mp", "sharded_rank_main.py")


def _gpus():
    try:
        import torch

        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def cases_for(world):
    """Cases every rank of which owns hypercubes: equal slabs, unequal slabs (planes not divisible by the world size) with a
    border in every dimension, a 2D f64 grid (header of 64-bit streams: two entries per word, odd counts padded), 1D."""
    return [f"float32:{16 * world},32,48",
            f"float32:{16 * (world + 1) + 5},37,41",
            f"float64:{64 * (world + 1)},200",
            f"float64:{4096 * (2 * world + 1) + 77}"]


def launch_ranks(world, out_dir, cases, backend, extra=(), timeout=600, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), RANK_MAIN, "--backend", backend, "--out", str(out_dir), *extra]
    for c in cases:
        cmd += ["--case", c]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    # (own process group: a hung collective must not outlive the test)
    proc = subprocess.Popen(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal

        os.killpg(proc.pid, signal.SIGKILL)
        out, _ = proc.communicate()
        pytest.fail(f"{world} ranks over {backend} did not finish in {timeout} s\n" + out[-3000:])
    assert proc.returncode == 0, out[-4000:]


def check_against_oracle(world, out_dir, cases):
    for i, case in enumerate(cases):
        dtype, extent, full = case_data(case)
        want = oracle.compress(full)
        shards = plan_shards(extent, world)
        parts = [np.load(os.path.join(out_dir, f"rank{r}_case{i}.npz")) for r in range(world)]
        assert all(bool(p["roundtrip"]) for p in parts), f"{case}: a rank's decompress did not reproduce its slab"
        for p in parts[1:]:
            assert np.array_equal(parts[0]["header"], p["header"]), f"{case}: every rank must hold the same global header"
        base = 0
        for p, s in zip(parts, shards):  # base_r = words of all lower ranks' hypercube runs (offset_header_gathered_kernel)
            assert int(p["base"]) == base, case
            base += len(p["body"]) - s.border
        got = assemble_stream(dtype, extent, parts[0]["header"], [p["body"] for p in parts], [len(p["body"]) for p in parts], shards)
        assert len(got) == len(want) and np.array_equal(got, want), f"{case}: assembled stream differs from the oracle's"


@pytest.mark.gpu
@pytest.mark.hardware_only
@pytest.mark.skipif(_gpus() < 2, reason="needs at least 2 GPUs (one process per GPU over RCCL)")
@pytest.mark.parametrize("mode", [[], ["--async-header-gather"], ["--overlap-exchange"], ["--native"], ["--native", "--overlap-exchange"]],
                         ids=["sync-header-gather", "async-header-gather", "overlap-exchange", "native-cpp-host", "native-cpp-host-overlap"])
def test_all_gpus_over_rccl_reproduce_the_single_stream(tmp_path, mode):
    world = _gpus()
    cases = cases_for(world)
    launch_ranks(world, tmp_path, cases, "nccl", extra=mode)
    check_against_oracle(world, tmp_path, cases)
