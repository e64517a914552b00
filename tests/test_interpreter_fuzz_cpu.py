"""A fixed slice of tools/fuzz_interpreter_vs_compiler.py inside the suite: random HIP kernels, compiled by hipcc for gfx950 and run
by the instruction-level interpreter, against the same programs compiled for the host.  The tool's long campaigns are recorded in
profiles/r06_code_object_rehearsal.txt; here a few dozen cases keep the interpreter (and its hazard / s_waitcnt checkers, which
compiler output must never trip) honest on every run."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists("/opt/rocm/lib/llvm/bin/clang++")), reason="needs hipcc and clang++")


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_ivc", os.path.join(ROOT, "tools", "fuzz_interpreter_vs_compiler.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("intrinsics,opt,first,model", [(False, "O3", 9100000, False), (True, "O3", 9200000, False), (True, "O1", 9300000, False),
                                                      (True, "O3", 9400000, True)])
def test_random_kernels_agree_with_their_host_build(tmp_path, intrinsics, opt, first, model):
    """model=True: a third leg -- the same device text on the functional model (tests/wavesim) under a random fiber schedule"""
    from tests import gfx950_exec as gx

    tool = _tool()
    tally = {}
    for seed in range(first, first + 10):
        status, info = tool.run_case(seed, str(tmp_path), opt, 28, gx, intrinsics, model)
        tally[status] = tally.get(status, 0) + 1
        # (the last two: the COMPILER's -- a load it sank past a barrier, its two instruction selectors disagreeing; tests/test_compiler_sink_audit.py)
        assert status in ("ok", "unknown-op", "unsupported", "compiler-sunk-load", "codegen-disagreement", "compiler-bitop3"), f"case {seed}: {status}: {info}"
    assert tally.get("ok", 0) >= 6, tally  # (the rest: kernels in which the compiler used an instruction the interpreter does not know)


def test_signed_bitfield_extract_past_bit_31_is_arithmetic():
    """What the fuzz found: V_BFE_I32 shifts its SIGNED source arithmetically, so a field that runs past bit 31 is filled with the
    sign (LLVM folds sbfe(x, off, n), off + n >= 32, to ashr(x, off)).  The product's kernels extract single bits only."""
    import numpy as np
    from tests import gfx950_exec as gx

    a = np.full(64, 0x80000000, dtype=np.uint32)
    got = gx._bfe_i32(a, np.uint32(31), np.uint32(15))
    assert (got == 0xFFFFFFFF).all()
    got = gx._bfe_i32(np.full(64, 0x40000000, dtype=np.uint32), np.uint32(30), np.uint32(15))
    assert (got == 1).all()
    got = gx._bfe_i32(np.full(64, 0x00000005, dtype=np.uint32), np.uint32(0), np.uint32(3))
    assert (got == 0xFFFFFFFD).all()
