"""The `-m gpu` suite, rehearsed: its test CODE (the oracle comparisons, the C-ABI calls, the torch plumbing) runs here on the
CPU with host tensors against the kernels' functional model (`pytest -m gpu --rehearse-on-model`, tests/conftest.py), so that
the one GPU run a round gets is spent on the hardware and not on a typo in a test.  Hardware-only cases (background
load, tools linked against the real library, process-isolated fault test) are skipped by the rehearsal.
This is not the GPU run and proves nothing about the MI355X."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_suite_passes_on_the_functional_model():
    cmd = [sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--rehearse-on-model", "-q", "-x", "-p", "no:cacheprovider"]
    try:  # (three test processes side by side when pytest-xdist is there: the full-size cases dominate and overlap)
        import xdist  # noqa: F401

        cmd += ["-n", "3"]
    except ImportError:
        pass
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 200, tail
    assert "failed" not in tail and "error" not in tail
