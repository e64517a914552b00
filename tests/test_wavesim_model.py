"""The wave64 functional model itself (test infrastructure, tests/wavesim): it must REJECT what the hardware would silently get
wrong.  wave_uniform / scalar_pointer are v_readfirstlane on the GPU -- lane 0's value for everybody -- so a claim that is false
for some lane has to abort the model run instead of passing with each lane's own value."""
import os
import signal
import subprocess

import pytest

from tests.wavesim import build as wbuild

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wavesim")


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    out = tmp_path_factory.mktemp("selftest") / "selftest_uniform"
    cmd = [wbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", HERE, os.path.join(HERE, "selftest_uniform.cc"),
           os.path.join(HERE, "wavesim.cc"), "-ldl", "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return str(out)


@pytest.mark.parametrize("mode", [0, 2, 3, 5])
def test_true_uniformity_claims_pass(selftest, mode):
    for schedule in ("", "reverse", "random:7"):
        r = subprocess.run([selftest, str(mode)], capture_output=True, text=True, env={**os.environ, "WAVESIM_SCHEDULE": schedule})
        assert r.returncode == 0 and r.stdout.startswith("ok"), (mode, schedule, r.stderr)


@pytest.mark.parametrize("mode", [1, 4])
def test_false_uniformity_claims_abort_with_the_call_site(selftest, mode):
    for schedule in ("", "reverse", "random:7"):
        r = subprocess.run([selftest, str(mode)], capture_output=True, text=True, env={**os.environ, "WAVESIM_SCHEDULE": schedule})
        assert r.returncode == -signal.SIGABRT, (mode, schedule, r.returncode, r.stdout)
        assert "claim is false" in r.stderr and "call site" in r.stderr
