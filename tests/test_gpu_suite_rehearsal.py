"""The `-m gpu` suite, rehearsed: its test CODE (the oracle comparisons, the C-ABI calls, the torch plumbing) runs here on the
CPU with host tensors against the kernels' functional model (`pytest -m gpu --rehearse-on-model`, tests/conftest.py), so that
the one GPU run a round gets is spent on the hardware and not on a typo in a test.  Hardware-only cases (background
load, tools linked against the real library, process-isolated fault test) are skipped by the rehearsal.
This is not the GPU run and proves nothing about the MI355X."""
import os
import re
import signal
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1800)  # (above the child's own limit below; pytest.ini's 600 s would kill this test first and orphan the child)
def test_gpu_suite_passes_on_the_functional_model():
    cmd = [sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--rehearse-on-model", "-q", "-x", "-p", "no:cacheprovider"]
    try:  # (test processes side by side when pytest-xdist is there: the full-size cases dominate and overlap -- 8 cores: 5 workers
        # 155 s against 204 s with 3; every worker's "GPU" is itself a handful of threads, so not one per core)
        import xdist  # noqa: F401

        cmd += ["-n", str(max(2, min(5, (os.cpu_count() or 4) - 3)))]
    except ImportError:
        pass
    # the child pytest (and its xdist workers) in a process group of its own, so that a timeout takes all of them down
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=1500)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail("the rehearsal of the GPU suite did not finish in 1500 s\n" + out[-2000:])
    r = subprocess.CompletedProcess(cmd, proc.returncode, out, err)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 200, tail
    assert "failed" not in tail and "error" not in tail
