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


def test_hardware_leg_plumbing_against_a_mock_runtime(tmp_path):
    """tools/fuzz_interpreter_vs_compiler.py --hardware (tests/test_zz_hip_fuzz_hardware.py on a GPU box) has never met a device: here its
    Python side -- buffers, the HIP_LAUNCH_PARAM `extra` array, the read-back, the comparison -- runs against a stand-in for libamdhip64
    whose hipModuleLaunchKernel unpacks the arguments as the runtime would and executes the code object on the interpreter."""
    import ctypes as C

    from tests import gfx950_exec as gx

    tool = _tool()

    class Tensor:
        def __init__(self, a):
            self.a = a.copy()

        def cuda(self):
            return self

        def cpu(self):
            return self

        def data_ptr(self):
            return self.a.ctypes.data

        def numpy(self):
            return self.a

    class Torch:
        class cuda:
            synchronize = staticmethod(lambda: None)

        from_numpy = staticmethod(Tensor)

    class Hip:
        def hipModuleLoad(self, ref, path):
            self.path = path.decode()
            return 0

        def hipModuleGetFunction(self, ref, mod, name):
            self.name = name.decode()
            return 0

        def hipModuleLaunchKernel(self, fn, gx_, gy, gz, bx, by, bz, shmem, stream, params, extra):
            assert params is None and (extra[0], extra[2], extra[4]) == (1, 2, 3)  # HIP_LAUNCH_PARAM_BUFFER_POINTER / _SIZE / _END
            size = C.cast(extra[3], C.POINTER(C.c_size_t))[0]
            gx.run_grid(gx.Kernel(gx.CodeObject(self.path), self.name), gx_, bx, shmem, C.string_at(extra[1], size), resident=2, quantum=400)
            return 0

        hipDeviceSynchronize = staticmethod(lambda: 0)
        hipModuleUnload = staticmethod(lambda mod: 0)

    hw = tool.Hardware.__new__(tool.Hardware)
    hw.torch, hw.hip = Torch, Hip()
    seen = set()
    for seed in (9500000, 9500001, 9500003):  # (one of them ends in a ticket loop: the tile region and the table travel too)
        status, _ = tool.run_case(seed, str(tmp_path), "O3", 28, gx, True, False, hw)
        seen.add(status)
    assert seen == {"ok"}, seen
