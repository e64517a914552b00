"""This image's LLVM can sink an LDS load past __syncthreads() (tools/audit_machine_sink.py: found by the differential fuzz of the
interpreter against the compiler).  Held here on every run:

  * the compiler's own account (MIR before / after its machine-sink pass) shows NO load leaving its block in any of the product's
    three device units -- so the product's code objects do not carry that race;
  * the reproducer is the positive control: the audit sees the sunk load, the interpreter runs the resulting code to a wrong
    answer under a skewed schedule, and with the pass switched off (-mllvm -disable-machine-sink) both are clean.  Should a later
    compiler stop sinking the load, the control asserts only that the program then runs right."""
import importlib.util
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _tool():
    spec = importlib.util.spec_from_file_location("audit_machine_sink", os.path.join(ROOT, "tools", "audit_machine_sink.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("unit", ["kernels_f32", "kernels_f64", "capi"])
def test_no_load_of_the_product_leaves_its_block(tmp_path, unit):
    tool = _tool()
    rows = tool.audit(os.path.join(ROOT, "ndzip_amd", "csrc", unit + ".hip"), tool.PRODUCT_FLAGS, str(tmp_path))
    assert len(rows) >= 4
    for name, nmoved, nloads, bad in rows:
        assert nloads == 0 and not bad, (name, nloads, bad)


def _expected(x):
    n = 512
    a, b = x[:n].astype(np.uint64), x[n:2 * n].astype(np.uint64)
    out = np.zeros(2 * n, dtype=np.uint32)
    for g0 in range(0, n, 256):
        t = np.arange(256)
        slot = np.zeros(256, dtype=np.uint64)
        slot[(t * 11 + 201) & 255] = (a[g0:g0 + 256] << np.uint64(32)) | b[g0:g0 + 256]
        first = slot[(t * 60 + 241) & 255]
        xx = a[g0:g0 + 256].astype(np.uint32).copy()
        bb = b[g0:g0 + 256].astype(np.uint32)
        for w0 in range(0, 256, 64):
            ballot = sum(int(bb[w0 + l] & 1) << l for l in range(64))
            for k in range(ballot & 3):
                xx[w0:w0 + 64] ^= bb[w0:w0 + 64] >> np.uint32(k)
        slot = np.zeros(256, dtype=np.uint64)
        slot[(t * 15 + 249) & 255] = xx.astype(np.uint64)
        second = slot[(t * 27 + 191) & 255]
        out[2 * g0:2 * g0 + 512:2] = ((first ^ (first >> np.uint64(32))).astype(np.uint32) + xx)
        out[2 * g0 + 1:2 * g0 + 512:2] = (second ^ (second >> np.uint64(32))).astype(np.uint32)
    return out


@pytest.mark.parametrize("sink", [True, False])
def test_reproducer_is_seen_by_the_audit_and_by_the_interpreter(tmp_path, sink):
    from tests import gfx950_exec as gx

    tool = _tool()
    src = tmp_path / "rep.hip"
    src.write_text(tool.REPRODUCER)
    flags = ["-O3"] + ([] if sink else ["-mllvm", "-disable-machine-sink"])
    sunk = sum(len(bad) for _, _, _, bad in tool.audit(str(src), flags, str(tmp_path))) if sink else 0  # (pass off: nothing to dump)
    co = tmp_path / "rep.hsaco"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "--genco", "--no-gpu-bundle-output", str(src), "-o", str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    x = np.random.default_rng(7).integers(0, 1 << 32, size=1024, dtype=np.uint64).astype(np.uint32)
    got = np.zeros(1024, dtype=np.uint32)
    k = gx.Kernel(gx.CodeObject(str(co)), "k_sink")
    assert not k.missing, k.missing
    gx.run_grid(k, 2, 256, 0, struct.pack("<QQ", x.ctypes.data, got.ctypes.data), resident=2, quantum=400)  # (one wavefront runs far ahead of the next)
    right = np.array_equal(got, _expected(x))
    if not sink:
        assert sunk == 0 and right
    elif sunk:
        assert not right, "the load was sunk past the barrier, yet the skewed schedule did not expose the race"
    else:
        assert right  # (a compiler that no longer sinks the load)


# ---- the second finding of the differential fuzz: SelectionDAG loses the high half of `x64 | ~zext(u32)` ---------------------------
# (AMD LLVM 22.0.0git of ROCm 7.2, hipcc -O1 .. -O3, gfx950 and gfx90a alike: when the ~zext value has a second use, the high half of
# the OR -- all ones, whatever x is, uniform or per lane -- is taken from a register that was never written; GlobalISel compiles the
# same IR correctly.)
# Nothing static can audit the product for a wrong instruction selection: that is what executing the BUILT code objects against the
# oracle is for (tests/test_gfx950_exec.py, the -m gpu suite rehearsed on the interpreter).  Here the 6-line kernel is the control
# that the interpreter tells the two builds apart.
OR_NOT_ZEXT = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void k_or(const uint32_t *in, uint64_t *out, uint64_t x) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint64_t n = ~(uint64_t) in[i];     // high half: all ones
    out[2 * i] = n;
    out[2 * i + 1] = x | n;                   // high half: all ones, whatever x is
}
"""


def test_selectiondag_or_of_not_zext_is_told_apart_from_globalisel(tmp_path):
    from tests import gfx950_exec as gx

    src = tmp_path / "or.hip"
    src.write_text(OR_NOT_ZEXT)
    x = np.random.default_rng(3).integers(0, 1 << 32, size=512, dtype=np.uint64).astype(np.uint32)
    xarg = 0x0123456789ABCDEF
    want = np.zeros(1024, dtype=np.uint64)
    want[0::2] = ~x.astype(np.uint64)
    want[1::2] = np.uint64(xarg) | ~x.astype(np.uint64)
    right = {}
    for name, flags in (("selectiondag", []), ("globalisel", ["-mllvm", "-global-isel", "-mllvm", "-global-isel-abort=2"])):
        co = tmp_path / f"{name}.hsaco"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--genco", "--no-gpu-bundle-output", *flags, str(src), "-o", str(co)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        k = gx.Kernel(gx.CodeObject(str(co)), "k_or")
        assert not k.missing, k.missing
        got = np.zeros(1024, dtype=np.uint64)
        gx.run_grid(k, 2, 256, 0, struct.pack("<QQQ", x.ctypes.data, got.ctypes.data, xarg), resident=2, quantum=400)
        right[name] = bool(np.array_equal(got, want))
        if not right[name]:
            assert np.array_equal(got[0::2], want[0::2]) and np.array_equal(got[1::2] & np.uint64(0xFFFFFFFF), want[1::2] & np.uint64(0xFFFFFFFF)), "only the high half of the OR is lost"
    assert right["globalisel"], right
    # (selectiondag: wrong with this image's compiler, right with a fixed one -- either way the interpreter has said which)


# ---- the third finding: a wrong V_BITOP3_B32 truth table when an inner bitwise value is reached twice from one root ----------------
# (gfx950 only -- the instruction is new there -- and in BOTH instruction selectors, which share the matcher; tools/audit_bitop3.py.)
# This one can be looked for in the product: its optimised IR must not contain the trigger shape.

def _bitop3_tool():
    spec = importlib.util.spec_from_file_location("audit_bitop3", os.path.join(ROOT, "tools", "audit_bitop3.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("unit", ["kernels_f32", "kernels_f64", "capi"])
def test_no_bitwise_expression_of_the_product_has_the_bitop3_trigger_shape(tmp_path, unit):
    tool = _bitop3_tool()
    hits, nroots = tool.audit(os.path.join(ROOT, "ndzip_amd", "csrc", unit + ".hip"), tool.PRODUCT_FLAGS, str(tmp_path))
    assert nroots >= 5 and not hits, hits[:5]


def test_bitop3_reproducer_is_flagged_and_its_table_is_known(tmp_path):
    tool = _bitop3_tool()
    hits, nroots = tool.scan_ir(tool.REPRODUCER_IR)
    assert nroots == 4 and [(h[1], h[2]) for h in hits] == [("%r", ["%t"])]
    # the table by the instruction's definition: S0 = 0xf0, S1 = 0xcc, S2 = 0xaa pushed through the expression
    A, b, X = 0xF0, 0xCC, 0xAA
    t = A & b
    assert ((t & X) | (t ^ b)) == tool.REPRODUCER_TABLE == 0x8C
    for extra in ((), ("-global-isel",)):
        got = tool.emitted_table(tool.REPRODUCER_IR, str(tmp_path), extra)
        assert got in (0x8C, 0xAC, None), hex(got)  # (0xac: this image's compiler, both selectors; 0x8c / no fusion: a fixed one)
    # a tree-shaped expression of the same size is fused correctly by this compiler: (A & X) | (b ^ X) has no shared inner value
    tree = "define i32 @f(i32 %A, i32 %b, i32 %X) {\n  %u = and i32 %A, %X\n  %v = xor i32 %b, %X\n  %r = or i32 %u, %v\n  ret i32 %r\n}\n"
    assert not tool.scan_ir(tree)[0]
    got = tool.emitted_table(tree, str(tmp_path))
    assert got in (None, (A & X) | (b ^ X)), hex(got)


# ---- the fourth finding: v_perm_b32 formation looks through `ashr (amdgcn.perm ...), 8k` past the end of the word -------------------
# (SelectionDAG, gfx950 and gfx90a; tools/audit_perm_sra.py.)  The bytes that should be copies of the sign come out as the low bytes.

def _perm_tool():
    spec = importlib.util.spec_from_file_location("audit_perm_sra", os.path.join(ROOT, "tools", "audit_perm_sra.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("unit", ["kernels_f32", "kernels_f64", "capi"])
def test_the_product_never_shifts_a_perm_result_arithmetically_by_whole_bytes(tmp_path, unit):
    tool = _perm_tool()
    hits, nperm, nshift = tool.audit(os.path.join(ROOT, "ndzip_amd", "csrc", unit + ".hip"), tool.PRODUCT_FLAGS, str(tmp_path))
    assert not hits, hits[:5]
    assert (nperm > 500 and nshift > 100) or unit == "capi"  # (the scan does see the builtin's calls and the `lshr 4` applied to their results)


def test_perm_under_an_arithmetic_shift_is_told_apart_from_globalisel(tmp_path):
    from tests import gfx950_exec as gx

    tool = _perm_tool()
    src = tmp_path / "perm.hip"
    src.write_text(tool.REPRODUCER)
    assert len(tool.audit(str(src), ["-O3"], str(tmp_path))[0]) == 1
    x = np.random.default_rng(5).integers(0, 1 << 32, size=1024, dtype=np.uint64).astype(np.uint32)
    d = x[512:]
    sign = np.where((d >> np.uint32(16)) & np.uint32(0x80), 0xFF, 0).astype(np.uint32)
    want = sign | (((d >> np.uint32(24)) & np.uint32(0xFF)) << np.uint32(8)) | (sign << np.uint32(16)) | (((d >> np.uint32(8)) & np.uint32(0xFF)) << np.uint32(24))
    right = {}
    for name, flags in (("selectiondag", []), ("globalisel", ["-mllvm", "-global-isel", "-mllvm", "-global-isel-abort=2"])):
        co = tmp_path / f"{name}.hsaco"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--genco", "--no-gpu-bundle-output", *flags, str(src), "-o", str(co)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        k = gx.Kernel(gx.CodeObject(str(co)), "k_perm")
        assert not k.missing, k.missing
        got = np.zeros(512, dtype=np.uint32)
        gx.run_grid(k, 2, 256, 0, struct.pack("<QQ", x.ctypes.data, got.ctypes.data), resident=2, quantum=400)
        right[name] = bool(np.array_equal(got, want))
        if not right[name]:  # only the two sign bytes are lost
            assert np.array_equal(got & np.uint32(0xFF00FF00), want & np.uint32(0xFF00FF00))
    assert right["globalisel"], right
