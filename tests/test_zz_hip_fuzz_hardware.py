"""The offline evidence of this repository rests on tests/gfx950_exec.py reading the ISA the way the silicon does.  The CPU suite holds
the interpreter against the COMPILER (tests/test_interpreter_fuzz_cpu.py: random HIP kernels, gfx950 code on the interpreter vs the
host build); here, on a GPU box, the same random kernels' code objects run on the DEVICE as well, and the device's words must be the
interpreter's -- every one, whatever the host build says (a disagreement of both with the host is the compiler's, see
tests/test_compiler_sink_audit.py).

hardware_only, and last in the session (tests/conftest.py): nothing rehearsed can be hidden by a surprise here.  A failure of the
plumbing (HIP module API through ctypes, never run before a GPU was reachable) is a skip with its message, not a verdict."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.hardware_only]


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_ivc", os.path.join(ROOT, "tools", "fuzz_interpreter_vs_compiler.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("intrinsics,opt,first", [(True, "O3", 9500000), (True, "O1", 9600000)])
def test_device_and_interpreter_agree_on_random_kernels(tmp_path, intrinsics, opt, first):
    from tests import gfx950_exec as gx

    tool = _tool()
    try:
        hw = tool.Hardware()
    except Exception as e:  # no device, no runtime library
        pytest.skip(f"hardware leg not available: {e}")
    tally = {}
    for seed in range(first, first + 12):
        try:
            status, info = tool.run_case(seed, str(tmp_path), opt, 28, gx, intrinsics, False, hw)
        except RuntimeError as e:
            if "HIP error" in str(e) or "hipModule" in str(e):
                pytest.skip(f"HIP module plumbing: {e}")
            raise
        tally[status] = tally.get(status, 0) + 1
        assert status != "HARDWARE-MISMATCH", f"case {seed}: the device and the interpreter differ: {info}"
        assert status in ("ok", "unknown-op", "unsupported", "compiler-sunk-load", "codegen-disagreement", "compiler-bitop3"), f"case {seed}: {status}: {info}"
    assert tally.get("ok", 0) >= 7, tally
