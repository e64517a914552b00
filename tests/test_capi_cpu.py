"""CPU suite: the C-ABI libraries load, export every symbol include/ndzip_hip.h / ndzip_hip_stages.h declare, its host-side sizing logic
matches the oracle, and -- without a GPU -- every compute entry point fails loudly (there is no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest

import ndzip_amd
from ndzip_amd import hip
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="ndzip_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"NDZIP_HIP_API[^;(]*?\b(ndzip_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(hip.EXPORTED_SYMBOLS) == declared, "hip.py symbol table out of sync with include/ndzip_hip.h"
    L = hip.lib()
    for name in declared:
        assert getattr(L, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ndzip_hip_\w+)", out))
    assert exported == set(declared), f"unexpected exports: {exported ^ set(declared)}"


def test_stage_hooks_live_in_a_library_of_their_own():
    """include/ndzip_hip_stages.h (the parity tests' single-hypercube stage entry point) is exported by libndzip_hip_stages.so and by
    nothing else: the product library holds neither the entry point nor a stage kernel."""
    declared = _declared_symbols("ndzip_hip_stages.h")
    assert sorted(hip.STAGE_SYMBOLS) == declared == ["ndzip_hip_debug_scratch_epoch_offset", "ndzip_hip_debug_stage", "ndzip_hip_stages_last_error"]
    out = subprocess.run(["nm", "-D", "--defined-only", hip.STAGES_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert set(re.findall(r" T (ndzip_hip_\w+)", out)) == set(declared)
    S = hip.stages_lib()
    for name in declared:
        assert getattr(S, name) is not None
    for path, want in ((hip.LIB_PATH, False), (hip.STAGES_LIB_PATH, True)):
        blob = open(path, "rb").read()
        assert (b"debug_stage_kernel" in blob) == want and (b"debug_lookback_kernel" in blob) == want, path
    blob = open(hip.STAGES_LIB_PATH, "rb").read()  # ... and the stage library holds no production kernel
    assert b"compress_kernel_db" not in blob and b"decompress_kernel" not in blob and b"compress_kernel_wide" not in blob


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sizing_matches_oracle(dtype):
    shapes = [(0,), (1,), (5,), (4096,), (4099,), (12305,), (64, 64), (70, 200), (200, 131), (1, 4096), (16, 16, 16), (17, 35, 33),
              (50, 37, 41), (3, 4, 15), (512, 512, 512), (8192, 8192), (2048, 1024, 1024)]
    for shape in shapes:
        assert ndzip_amd.compressed_length_bound(dtype, shape) == oracle.compressed_length_bound(dtype, shape)
        assert ndzip_amd.num_hypercubes(shape) == oracle.num_hypercubes(shape)
    for nhc in (0, 1, 2, 3, 32768):
        assert ndzip_amd.header_words(dtype, nhc) == (nhc if dtype == np.float32 else (nhc + 1) // 2)


def test_compressor_requirements_semantics():
    """ndzip::compressor_requirements: max hypercubes over extents of one dimensionality (common.cc:8-28)"""
    req = ndzip_amd.CompressorRequirements((64 * 3, 64 * 2), (64, 64 * 5))
    assert req.dims == 2 and req.max_num_hypercubes == 6
    with pytest.raises(RuntimeError, match="-dimensional"):
        req.include((4096,))


def test_invalid_arguments_are_rejected_on_the_host():
    with pytest.raises(ndzip_amd.NdzipHipError):
        ndzip_amd.compressed_length_bound(np.float32, (1, 2, 3, 4))
    with pytest.raises(TypeError):
        ndzip_amd.compressed_length_bound(np.int32, (16,))
    with pytest.raises(ndzip_amd.NdzipHipError, match="[Dd]imensionality"):
        ndzip_amd.make_hip_offloader(np.float32, 4)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(ndzip_amd.NdzipHipError) as e:
        ndzip_amd.make_hip_offloader(np.float32, 1).compress(np.zeros(4096, np.float32))
    assert e.value.status == hip.ERR_NO_DEVICE
    with pytest.raises(ndzip_amd.NdzipHipError) as e:
        ndzip_amd.make_hip_compressor(np.float32, ndzip_amd.CompressorRequirements((4096,)))
    assert e.value.status == hip.ERR_NO_DEVICE
    with pytest.raises(ndzip_amd.NdzipHipError):
        ndzip_amd.device_info()


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under ndzip_amd/ or bench.py's product path may import it."""
    for base, _, files in os.walk(os.path.join(ROOT, "ndzip_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inl", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "libndzip_oracle", "ndzip_oracle_", "ndzip_ref_", "oracle/"):
                    assert needle not in text, f"{f} references the oracle ({needle})"


def test_compress_and_decompress_kernels_do_not_spill():
    """Register budget of the hot kernels (hipcc's resource remarks of the in-tree build).  Scratch in a compress kernel is a
    performance bug, not just slowness: a scratch reload is a vector-memory load, it retires in order behind every prefetch
    load of the wavefront and so waits out a full HBM round trip."""
    from ndzip_amd import build

    build.build()
    res = build.kernel_resources()
    if not res:  # objects older than the resource capture: rebuild once
        build.build(force=True)
        res = build.kernel_resources()
    hot = {k: v for k, v in res.items() if "compress_kernel" in k}  # compress_kernel, compress_kernel_db, decompress_kernel
    assert len(hot) == 31, sorted(res)  # f32 db (7: 6 + the paired 3D variant) + f64 wide (6) + decompress (12) + f64 wide decompress (6)
    for name, r in hot.items():
        assert r["scratch"] == 0, f"{name} spills {r['scratch']} bytes per lane"
    # every compress kernel runs 4 workgroups per CU: 128 VGPRs at most, still no scratch (round 2: 143-168 VGPRs, 3 per CU)
    # the 256-work-item f64 decoder: 4 workgroups of 4 wavefronts per CU by its LDS (35-36 KB), so 4 wavefronts per SIMD must fit
    # the registers with room to spare (the 128-work-item one: 146 VGPRs, 3 by registers and 2 by LDS)
    wide_dec = {k: v for k, v in hot.items() if "decompress_kernel_wide" in k}
    assert len(wide_dec) == 6 and all(r["occupancy"] >= 4 and r["vgprs"] <= 96 for r in wide_dec.values()), wide_dec
    comp = {k: v for k, v in hot.items() if "decompress" not in k}
    assert len(comp) == 13 and all(r["occupancy"] == 4 and r["vgprs"] <= 128 for r in comp.values()), comp
    # SGPR spills are VGPR-lane traffic (v_readlane + hazard nops) inside the persistent loop: 45-49 before round 3
    assert all(r["sgpr_spill"] <= 24 for r in comp.values()), comp
    dec = [v for k, v in hot.items() if "decompress_kernelIf" in k]
    assert dec and all(r["occupancy"] >= 5 for r in dec)                 # LDS admits 4 workgroups of 4 wavefronts per CU anyway


def test_pipelined_offloader_validates_arguments_before_touching_the_device():
    """ndzip_hip_offloader_create: argument errors are reported as such (not as a missing device), and without a device the call
    fails loudly -- there is no CPU fallback behind the persistent host-pointer interface either."""
    import ctypes as C

    L = hip.lib()
    h = C.c_void_p()
    ext = (C.c_uint32 * 3)(64, 64, 64)
    assert L.ndzip_hip_offloader_create(hip.F32, 3, ext, 2, None) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_create(7, 3, ext, 2, C.byref(h)) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_create(hip.F32, 4, ext, 2, C.byref(h)) == hip.ERR_INVALID_ARGUMENT
    assert b"Invalid dimensionality" in L.ndzip_hip_last_error()
    assert L.ndzip_hip_offloader_create(hip.F32, 3, ext, 0, C.byref(h)) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_create(hip.F32, 3, ext, 17, C.byref(h)) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_create(hip.F32, 3, None, 2, C.byref(h)) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_wait(None, 0, None, None) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_submit_compress(None, 0, ext, None, None) == hip.ERR_INVALID_ARGUMENT
    assert L.ndzip_hip_offloader_destroy(None) == hip.OK and L.ndzip_hip_host_free(None) == hip.OK
    if not os.path.exists("/dev/kfd"):
        assert L.ndzip_hip_offloader_create(hip.F32, 3, ext, 2, C.byref(h)) == hip.ERR_NO_DEVICE and not h.value
        p = C.c_void_p()
        assert L.ndzip_hip_host_alloc(4096, C.byref(p)) == hip.ERR_NO_DEVICE and not p.value
        with pytest.raises(ndzip_amd.NdzipHipError, match="no CPU fallback"):
            ndzip_amd.HipPipelinedOffloader(np.float32, (64, 64, 64))


def test_production_library_reads_no_experiment_knobs():
    """The only environment variable the shipped library looks at is NDZIP_VERBOSE (the reference's own tracing switch,
    src/ndzip/common.hh:630-633): a stray variable on a benchmark box must not change which kernel runs.  The experiment
    switches of tools/ are not in the product sources at all (tools/experiments/lab_scaffolding.patch)."""
    import re

    with open(hip.LIB_PATH, "rb") as f:
        blob = f.read()
    names = set(re.findall(rb"\x00(NDZIP_[A-Z0-9_]{3,})\x00", blob))  # whole C strings that look like a variable name
    assert names == {b"NDZIP_VERBOSE"}, names
    csrc = os.path.join(ROOT, "ndzip_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".inl")):
            text = open(os.path.join(csrc, f)).read()
            assert "NDZIP_EXP" not in text and "NDZIP_HIP_EXP" not in text, f"lab switch in product source {f}"


def test_product_package_never_touches_the_wave_model():
    """tests/wavesim (the CPU functional model of the kernels) is test infrastructure like the oracle: nothing under
    ndzip_amd/, include/ or the entry points may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base, _, files in os.walk(os.path.join(root, "ndzip_amd")):
        if "_build" in base or "__pycache__" in base or "_variants" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inl", ".cc", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if f == "gfx950_lds.hpp":  # its header comment says why the file exists
                    continue
                assert "wavesim" not in text, os.path.join(base, f)
    for f in ("bench.py", "__graft_entry__.py", os.path.join("include", "ndzip_hip.h"), os.path.join("include", "ndzip_hip.hh")):
        assert "wavesim" not in open(os.path.join(root, f)).read(), f


def test_lab_patch_still_applies():
    """tools/experiments/lab_scaffolding.patch (ablation flags, knobs, phase timers, alternative orderings: what
    tools/build_variant.sh --lab compiles) must keep applying to the product sources it was cut from."""
    import shutil
    import subprocess
    import tempfile

    csrc = os.path.join(ROOT, "ndzip_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(csrc):
            if f.endswith((".hip", ".hpp", ".inl")):
                shutil.copy(os.path.join(csrc, f), tmp)
        with open(os.path.join(ROOT, "tools", "experiments", "lab_scaffolding.patch")) as patch:
            r = subprocess.run(["patch", "-p1", "--dry-run", "-d", tmp], stdin=patch, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_shipped_kernels_use_no_flat_or_scratch_memory(tmp_path):
    """Disassembly of the gfx950 code objects inside the built library: every LDS access is a ds_ instruction and every global one
    a global_ instruction (an address computed through a generic pointer would show up as flat_*, a register spill as scratch_*),
    the read-once input carries the nt policy, and the look-back's descriptor traffic is write-through / system-coherent (sc1)."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    lib = shutil.copy(hip.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, capture_output=True, text=True, check=True)
    objects = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(objects) >= 3, os.listdir(tmp_path)
    text = ""
    for f in objects:
        text += subprocess.run([objdump, "-d", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
    ops = [line.split("\t")[1].split()[0] for line in text.splitlines() if line.startswith("\t") and len(line.split("\t")) > 1 and line.split("\t")[1].strip()]
    assert len(ops) > 20000
    assert not [o for o in ops if o.startswith(("flat_", "scratch_"))]
    assert sum(o.startswith("ds_") for o in ops) > 1000 and sum(o.startswith("global_load") for o in ops) > 300
    assert text.count(" nt") > 100 and text.count(" sc1") > 50


def test_exec_masked_plane_compaction_has_the_shape_it_was_written_in(tmp_path):
    """The branch-free plane compaction of the f32 compress kernels is inline assembly that changes EXEC (gfx950_lds.hpp:
    lds_append_nonzero).  In the disassembly of the built library every v_cmpx must be followed by its store, its address
    increment and the restore of EXEC from the SGPR pair that EXEC was saved to at the head of the sequence -- nothing the compiler
    scheduled may sit inside a masked stretch -- and every sequence of 32 must end in the s_nop that covers the DPP / v_readlane
    hazard after a VALU write of EXEC."""
    import re
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    from tests import gfx9_interp as gi

    # the product library (7 instantiations of compress_kernel_db, loop and drain) and the stage library (the f32 encode stage
    # kernels: 3 dims x aligned / unaligned), 32 planes per sequence
    ins = []
    for k, path in enumerate((hip.LIB_PATH, hip.STAGES_LIB_PATH)):
        os.makedirs(tmp_path / str(k))
        ins += gi.disassemble(path, str(tmp_path / str(k)))
    at = [i for i, s in enumerate(ins) if s.startswith("v_cmpx_ne_u32")]
    assert len(at) % 32 == 0 and len(at) >= (14 + 6) * 32, len(at)
    for n, i in enumerate(at):
        m = re.match(r"v_cmpx_ne_u32_e32 vcc, 0, (v\d+)$", ins[i])
        assert m, ins[i]
        word = m.group(1)
        st = re.match(rf"ds_write_b32 (v\d+), {word}$", ins[i + 1])
        assert st, ins[i:i + 4]
        assert ins[i + 2] == f"v_add_u32_e32 {st.group(1)}, 4, {st.group(1)}", ins[i:i + 4]
        rs = re.match(r"s_mov_b64 exec, (s\[\d+:\d+\])$", ins[i + 3])
        assert rs, ins[i:i + 4]
        if n % 32 == 0:  # head of a sequence: EXEC was saved to that very pair just before (hazard nops may sit in between)
            before = [s for s in ins[max(0, i - 4):i] if s == f"s_mov_b64 {rs.group(1)}, exec"]
            assert before, ins[max(0, i - 4):i + 4]
            saved = rs.group(1)
        assert rs.group(1) == saved
        if n % 8 == 7:   # end of an asm statement
            assert ins[i + 4] == "s_nop 1", ins[i:i + 6]


def test_exec_masked_plane_compaction_executes_correctly_under_any_entry_mask(tmp_path):
    """The model replaces lds_append_nonzero by a C++ loop, so nothing in the CPU suite ran the assembly itself.  This does: every
    32-plane sequence found in the BUILT code object (from the save of EXEC to the last s_nop) is interpreted for 64 lanes
    (tests/gfx9_interp.py: v_cmpx_ne_u32 / ds_write_b32 / v_add_u32 / s_mov_b64 / s_nop, anything else inside is an error) under
    random entry EXEC masks -- the sequence is entered from a per-lane branch -- and random plane words, and compared with the
    loop it replaces: LDS contents, the address behind the last word, untouched inactive lanes, EXEC restored."""
    import re

    import numpy as np

    from tests import gfx9_interp as gi

    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    ins = []
    for k, path in enumerate((hip.LIB_PATH, hip.STAGES_LIB_PATH)):  # (product kernels and the f32 encode stage kernels)
        os.makedirs(tmp_path / str(k))
        ins += gi.disassemble(path, str(tmp_path / str(k)))
    heads = []
    for i, s in enumerate(ins):
        m = re.fullmatch(r"s_mov_b64 (s\[\d+:\d+\]), exec", s)
        if m and any(x.startswith("v_cmpx_ne_u32") for x in ins[i + 1:i + 4]):
            heads.append((i, m.group(1)))
    assert len(heads) >= 20, len(heads)  # 7 compress_kernel_db instantiations (loop + drain) + 6 f32 encode stage kernels
    rng = np.random.default_rng(20260929)
    for n, (start, saved) in enumerate(heads):
        # the sequence: up to and including the 4th "s_nop 1" that closes an asm statement
        end, closes = start, 0
        while closes < 4:
            end += 1
            if ins[end] == "s_nop 1" and ins[end - 1].startswith("s_mov_b64 exec"):
                closes += 1
        seq = ins[start:end + 1]
        words = [re.fullmatch(r"v_cmpx_ne_u32_e32 vcc, 0, (v\d+)", s).group(1) for s in seq if s.startswith("v_cmpx")]
        addr = re.fullmatch(r"ds_write_b32 (v\d+), v\d+", next(s for s in seq if s.startswith("ds_write"))).group(1)
        assert len(words) == 32 and len(set(words)) == 32 and addr not in words
        for case in range(6 if n < 3 else 2):
            w = gi.Wave()
            entry = [gi.MASK64, 1, 0, 0xAAAAAAAA55555555][case] if case < 4 else int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
            w.exec = entry
            density = [0.6, 1.0, 0.5, 0.0, 0.3, 0.9][case % 6]
            planes = rng.integers(1, 1 << 32, size=(32, 64), dtype=np.uint64).astype(np.uint32)
            planes[rng.random((32, 64)) >= density] = 0
            for i, reg in enumerate(words):
                w.v[reg] = planes[i].copy()
            a0 = (np.arange(64, dtype=np.uint32) * 160 + rng.integers(0, 8, size=64).astype(np.uint32) * 4).astype(np.uint32)
            w.v[addr] = a0.copy()
            for s in seq:
                w.step(s)
            assert w.exec == entry, (n, case, hex(w.exec), hex(entry))
            want = np.zeros_like(w.lds)
            touched = np.zeros_like(w.lds_written)
            for lane in range(64):
                a = int(a0[lane])
                if (entry >> lane) & 1:
                    for i in range(32):
                        if planes[i, lane]:
                            want[a // 4] = planes[i, lane]
                            touched[a // 4] = True
                            a += 4
                assert int(w.v[addr][lane]) == a, (n, case, lane)
                for i, reg in enumerate(words):
                    assert int(w.v[reg][lane]) == int(planes[i, lane])
            assert np.array_equal(w.lds_written, touched) and np.array_equal(w.lds, want), (n, case)
