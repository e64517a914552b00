"""The instruction-level interpreter (tests/gfx950_exec.py) is what lets this suite say "the BUILT gfx950 code objects reproduce the
oracle" without a GPU -- so its own semantics need an authority that is not its author.  tests/test_isa_primitives_crosscheck.py
holds single instructions and small functions against LLVM; here whole KERNELS that are not the product's: plain HIP C++ whose meaning
the language and the HIP programming model define -- wavefront shuffles, ballots and votes, divergent loops with early exits and a
switch, LDS exchange through barriers with 4- and 16-byte accesses, 64-bit multiplies, and a divergent atomicAdd that LLVM's atomic
optimizer turns into a wave scan (once with its DPP strategy: row shifts, row broadcasts, whole-wave mode) -- compiled by hipcc for
gfx950 and executed by the interpreter.  The expected values are computed here from the source's meaning, in numpy.

What hipcc chooses to emit (ds_bpermute address arithmetic, s_and_saveexec / s_andn2_saveexec nests, s_or_saveexec whole-wave
sections, v_readlane, v_add_u32_dpp ... row_bcast:31 row_mask:0xc) is the compiler's business; that the interpreter gets the
source's answer out of it is the check."""
import os
import struct
import subprocess

import numpy as np
import pytest

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")

SOURCE = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void k_lanes(const uint32_t *in, uint32_t *out, unsigned long long *masks) {
    const int t = threadIdx.x;
    const uint32_t v = in[blockIdx.x * blockDim.x + t];
    uint32_t *o = out + (blockIdx.x * blockDim.x + t) * 16;
    o[0] = __shfl_xor(v, 1);
    o[1] = __shfl_xor(v, 2);
    o[2] = __shfl_xor(v, 16);
    o[3] = __shfl_xor(v, 32);
    o[4] = __shfl_up(v, 1);
    o[5] = __shfl_up(v, 5);
    o[6] = __shfl_down(v, 3);
    o[7] = __shfl(v, 7);
    o[8] = __shfl(v, (t * 7 + 3) & 63);
    o[9] = __shfl_xor(v, 4, 16);
    o[10] = __shfl_up(v, 2, 32);
    o[11] = __popcll(__ballot(v & 1));
    o[12] = __any(v > 0xfffffff0u) + 2 * __all(v != 0x12345678u);
    o[13] = __lane_id();
    o[14] = __builtin_amdgcn_readfirstlane(v);
    o[15] = __builtin_amdgcn_readlane(v, 37);
    const unsigned long long m = __ballot((v >> 3) & 1);   // (taken by all lanes: a ballot counts the ACTIVE lanes only)
    if (t == 0) masks[blockIdx.x] = m;
}
extern "C" __global__ void k_flow(const uint32_t *in, uint32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = in[i], acc = 0;
    for (uint32_t k = 0; k < (x & 31u); ++k) {
        if ((x >> k) & 1u) acc += k * 3u + 1u; else acc ^= (x >> (k & 7u));
        if (acc > 0xf0000000u) break;
    }
    switch (x & 3u) {
        case 0: acc += 11; break;
        case 1: acc = acc * 5u + (x >> 7); break;
        case 2: if (x & 4u) { acc -= 9; } else { acc |= 0x100u; } break;
        default: acc = ~acc;
    }
    uint64_t w = (uint64_t) x * 0x9e3779b97f4a7c15ull + acc;
    w ^= w >> 29;
    out[i] = (uint32_t) w ^ (uint32_t) (w >> 32);
}
extern "C" __global__ void k_lds(const uint32_t *in, uint32_t *out, uint32_t *counters) {
    __shared__ uint32_t a[256 + 8];
    __shared__ uint4 q[64];
    __shared__ uint32_t hist[8];
    const int t = threadIdx.x;
    a[t] = in[blockIdx.x * 256 + t];
    __syncthreads();
    const uint32_t r = a[255 - t] + a[(t * 17) & 255];
    if (t < 8) {
        uint32_t c = 0;
        for (int k = 0; k < 256; ++k) c += ((a[255 - k] + a[(k * 17) & 255]) & 7u) == (uint32_t) t;
        hist[t] = c;
    }
    __syncthreads();
    if (t < 64) q[t] = make_uint4(a[4 * t], a[4 * t + 1] ^ r, a[4 * t + 2], a[4 * t + 3] + t);
    __syncthreads();
    const uint4 z = q[(t * 5) & 63];
    out[blockIdx.x * 256 + t] = r ^ z.x ^ (z.y << 1) ^ (z.z >> 1) ^ z.w ^ hist[t & 7];
    if (t < 8) atomicAdd(&counters[t], hist[t]);
}
extern "C" __global__ void k_scan(uint32_t *total, const uint32_t *in, uint32_t *out) {
    const int t = threadIdx.x;
    const uint32_t v = in[t];
    out[t] = 0xffffffffu;
    if ((t & 3) != 1 && t != 40) out[t] = atomicAdd(total, v);
}
"""


@pytest.fixture(scope="module")
def objects(tmp_path_factory):
    from tests import gfx950_exec as gx

    d = tmp_path_factory.mktemp("compiler")
    (d / "t.hip").write_text(SOURCE)
    out = {}
    for name, flags in (("default", []), ("dpp", ["-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"])):
        co = d / f"{name}.hsaco"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--genco", "--no-gpu-bundle-output", *flags, str(d / "t.hip"), "-o", str(co)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = gx.CodeObject(str(co))
    return out


def _run(co, kernel, grid, block, *args, lds=0):
    from tests import gfx950_exec as gx

    k = gx.Kernel(co, kernel)
    assert not k.missing, f"{kernel}: the compiler used {k.missing}, which the interpreter does not know (change the SOURCE, not the interpreter)"
    packed = b"".join(struct.pack("<Q", a.ctypes.data) if isinstance(a, np.ndarray) else struct.pack("<i", a) for a in args)
    gx.run_grid(k, grid, block, lds, packed, resident=4, quantum=500)
    return {x.op for x in k.code.values()}


def test_wavefront_shuffles_ballots_and_votes(objects):
    rng = np.random.default_rng(1)
    grid, block = 2, 128
    v = rng.integers(0, 1 << 32, size=grid * block, dtype=np.uint64).astype(np.uint32)
    v[5] = 0xFFFFFFF5  # (one wavefront has a lane for which __any is true)
    out = np.zeros((grid * block, 16), dtype=np.uint32)
    masks = np.zeros(grid, dtype=np.uint64)
    ops = _run(objects["default"], "k_lanes", grid, block, v, out, masks)
    assert "ds_bpermute_b32" in ops and "v_readlane_b32" in ops
    for w0 in range(0, grid * block, 64):  # per wavefront
        x = v[w0:w0 + 64]
        lane = np.arange(64)
        want = np.zeros((64, 16), dtype=np.uint32)
        for col, m in ((0, 1), (1, 2), (2, 16), (3, 32)):
            want[:, col] = x[lane ^ m]
        want[:, 4] = x[np.where(lane >= 1, lane - 1, lane)]
        want[:, 5] = x[np.where(lane >= 5, lane - 5, lane)]
        want[:, 6] = x[np.where(lane + 3 < 64, lane + 3, lane)]
        want[:, 7] = x[7]
        want[:, 8] = x[((w0 % block + lane) * 7 + 3) & 63]
        want[:, 9] = x[lane ^ 4]                                           # width 16: the partner stays inside the 16-lane segment
        want[:, 10] = x[np.where((lane & 31) >= 2, lane - 2, lane)]         # width 32: no value crosses a 32-lane segment
        want[:, 11] = int((x & 1).sum())
        want[:, 12] = int((x > 0xFFFFFFF0).any()) + 2 * int((x != 0x12345678).all())
        want[:, 13] = lane
        want[:, 14] = x[0]
        want[:, 15] = x[37]
        assert np.array_equal(out[w0:w0 + 64], want), np.argwhere(out[w0:w0 + 64] != want)[:5]
    for b in range(grid):
        bits = (v[b * block:b * block + 64] >> 3) & 1
        assert int(masks[b]) == sum(int(bit) << k for k, bit in enumerate(bits))


def _flow(x):
    M = 0xFFFFFFFF
    acc = 0
    for k in range(x & 31):
        if (x >> k) & 1:
            acc = (acc + k * 3 + 1) & M
        else:
            acc ^= x >> (k & 7)
        if acc > 0xF0000000:
            break
    c = x & 3
    if c == 0:
        acc = (acc + 11) & M
    elif c == 1:
        acc = (acc * 5 + (x >> 7)) & M
    elif c == 2:
        acc = (acc - 9) & M if x & 4 else acc | 0x100
    else:
        acc = ~acc & M
    w = (x * 0x9E3779B97F4A7C15 + acc) & ((1 << 64) - 1)
    w ^= w >> 29
    return (w ^ (w >> 32)) & M


def test_divergent_loops_early_exits_and_a_switch(objects):
    rng = np.random.default_rng(2)
    grid, block, n = 3, 128, 3 * 128 - 37          # the last 37 work-items leave at once: partial EXEC from the first instruction on
    x = rng.integers(0, 1 << 32, size=grid * block, dtype=np.uint64).astype(np.uint32)
    x[:8] = [0, 1, 2, 3, 0xFFFFFFFF, 0xFFFFFFFE, 0x8000001F, 0x7FFFFFFD]
    x[8:40] |= 0xF8000000                          # large accumulators: the break fires for some
    out = np.full(grid * block, 0xABABABAB, dtype=np.uint32)
    ops = _run(objects["default"], "k_flow", grid, block, x, out, n)
    assert any(o.startswith("s_cbranch_exec") for o in ops) and any("saveexec" in o for o in ops)
    want = np.array([_flow(int(v)) for v in x[:n]], dtype=np.uint32)
    assert np.array_equal(out[:n], want), np.flatnonzero(out[:n] != want)[:8]
    assert (out[n:] == 0xABABABAB).all(), "work-items that returned early wrote something"


def test_lds_exchange_through_barriers(objects):
    rng = np.random.default_rng(3)
    grid = 2
    a_all = rng.integers(0, 1 << 32, size=grid * 256, dtype=np.uint64).astype(np.uint32)
    out = np.zeros(grid * 256, dtype=np.uint32)
    counters = np.array([5, 0, 0, 7, 0, 0, 0, 1], dtype=np.uint32)
    start = counters.copy()
    ops = _run(objects["default"], "k_lds", grid, 256, a_all, out, counters)
    assert "s_barrier" in ops and any(o.startswith("ds_read_b128") or o.startswith("ds_write_b128") for o in ops)
    t = np.arange(256)
    total_hist = np.zeros(8, dtype=np.uint32)
    for b in range(grid):
        a = a_all[b * 256:(b + 1) * 256]
        r = a[255 - t] + a[(t * 17) & 255]
        hist = np.bincount(r & 7, minlength=8).astype(np.uint32)
        total_hist += hist
        q = np.stack([a[0::4], a[1::4] ^ r[:64], a[2::4], a[3::4] + np.arange(64, dtype=np.uint32)], axis=1)
        z = q[(t * 5) & 63]
        want = r ^ z[:, 0] ^ (z[:, 1] << np.uint32(1)) ^ (z[:, 2] >> np.uint32(1)) ^ z[:, 3] ^ hist[t & 7]
        assert np.array_equal(out[b * 256:(b + 1) * 256], want.astype(np.uint32))
    assert np.array_equal(counters, start + total_hist)


@pytest.mark.parametrize("strategy", ["default", "dpp"])
def test_the_wave_scan_llvm_builds_for_a_divergent_atomic(objects, strategy):
    """atomicAdd(total, v) in a divergent branch: one lane adds the wavefront's sum, lane i receives old + the sum of the values of the
    ACTIVE lanes below it.  With the DPP strategy that exclusive scan is row_shr 1/2/4/8 + row_bcast:15 + row_bcast:31 additions in
    whole-wave mode with the inactive lanes set to 0 -- the product's own wave scans are these six steps (tests/test_isa_dpp_crosscheck.py
    compares the control words; this executes LLVM's use of them and checks the sums)."""
    rng = np.random.default_rng(4)
    v = rng.integers(0, 1 << 20, size=64, dtype=np.uint64).astype(np.uint32)
    total = np.array([1000], dtype=np.uint32)
    out = np.zeros(64, dtype=np.uint32)
    ops = _run(objects[strategy], "k_scan", 1, 64, total, v, out)
    if strategy == "dpp":
        assert "v_add_u32_dpp" in ops and "s_or_saveexec_b64" in ops, sorted(o for o in ops if "dpp" in o or "exec" in o)
    t = np.arange(64)
    active = ((t & 3) != 1) & (t != 40)
    prefix = np.concatenate([[0], np.cumsum(np.where(active, v, 0).astype(np.uint64))[:-1]]).astype(np.uint32)
    want = np.where(active, np.uint32(1000) + prefix, np.uint32(0xFFFFFFFF))
    assert np.array_equal(out, want), np.flatnonzero(out != want)[:8]
    assert int(total[0]) == 1000 + int(v[active].sum())


# ---- the five hand-written assembly blocks: assembled form == hipcc's compilation of the C++ fallback == the definition --------------
#
# ndzip_amd/csrc/gfx950_lds.hpp holds each block twice: as gfx950 assembly and, behind -DNDZIP_NO_EXEC_ASM, as plain C++ /
# __builtin_amdgcn_update_dpp that means the same; tests/wavesim/gfx950_lds.hpp holds a THIRD version, the one the functional model runs
# (assembly has no host meaning).  All three are compiled into kernels that do nothing else -- the first two by hipcc and executed by the
# interpreter, the third by the host compiler on the model -- run on the same random data, and held against what the operation IS (a
# 64-bit prefix sum, a pair exchange, a compaction) in numpy.  The interpreter's reading of the opcodes the FALLBACK compiles to is what the tests above and
# tests/test_isa_primitives_crosscheck.py pin to LLVM; the assembly has to land on the same numbers.

ASM_SOURCE = r"""
#include "gfx950_lds.hpp"
namespace ndzip_hip {   // (like the product's kernels: `extern __shared__ char smem[]` is the workgroup's dynamic LDS, on the GPU and on the model)
template<int D> __device__ void row_kernel(const uint32_t *in, uint32_t *out) {
    uint32_t lo[8], hi[8];
    for (int j = 0; j < 8; ++j) { lo[j] = in[threadIdx.x * 16 + j]; hi[j] = in[threadIdx.x * 16 + 8 + j]; }
    row_scan_step64<D>(lo, hi);
    for (int j = 0; j < 8; ++j) { out[threadIdx.x * 16 + j] = lo[j]; out[threadIdx.x * 16 + 8 + j] = hi[j]; }
}
extern "C" __global__ void k_row1(const uint32_t *in, uint32_t *out) { row_kernel<1>(in, out); }
extern "C" __global__ void k_row2(const uint32_t *in, uint32_t *out) { row_kernel<2>(in, out); }
extern "C" __global__ void k_row4(const uint32_t *in, uint32_t *out) { row_kernel<4>(in, out); }
extern "C" __global__ void k_row8(const uint32_t *in, uint32_t *out) { row_kernel<8>(in, out); }
extern "C" __global__ void k_scan64(const uint32_t *in, uint32_t *out) {
    uint32_t lo = in[threadIdx.x * 2], hi = in[threadIdx.x * 2 + 1];
    wave_inclusive_scan64(lo, hi);
    out[threadIdx.x * 2] = lo; out[threadIdx.x * 2 + 1] = hi;
}
extern "C" __global__ void k_pair(const uint32_t *in, uint32_t *out) {
    uint32_t a[4], b[4], lo[4], hi[4];
    for (int j = 0; j < 4; ++j) { a[j] = in[threadIdx.x * 8 + j]; b[j] = in[threadIdx.x * 8 + 4 + j]; }
    pair_exchange_select4(threadIdx.x & 1u, a, b, lo, hi);
    for (int j = 0; j < 4; ++j) { out[threadIdx.x * 8 + j] = lo[j]; out[threadIdx.x * 8 + 4 + j] = hi[j]; }
}
// every lane compacts its 32 words into its own 128-byte region of LDS (zeroed first); lanes with skip[t] sit the compaction out, so
// the block is entered under a PARTIAL exec mask; then every lane copies its region and the end address it got back out
extern "C" __global__ void k_append(const uint32_t *in, uint32_t *out, uint32_t *ends, const uint32_t *skip) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(smem);   // 64 x 32 words
    const int t = threadIdx.x;
    uint32_t w[32];
    for (int j = 0; j < 32; ++j) { w[j] = in[t * 32 + j]; lds[t * 32 + j] = 0; }
    __syncthreads();
    uint32_t end = 0xffffffffu;
    if (!skip[t]) end = lds_append_nonzero(lds_address(lds) + t * 128, w) - (lds_address(lds) + t * 128);
    lds_append_complete();
    __syncthreads();
    for (int j = 0; j < 32; ++j) out[t * 32 + j] = lds[t * 32 + j];
    ends[t] = end;
}
// the 64-bit profiles' compaction: dword w[i] kept where bit 31 - i of flags is set, stored at the XOR-swizzled address, the running
// address advancing by 8; a lane's region is 256 bytes (the swizzle stays inside a 128-byte block)
extern "C" __global__ void k_append64(const uint32_t *in, const uint32_t *flags, uint32_t *out, const uint32_t *skip) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(smem);   // 64 x 64 words
    const int t = threadIdx.x;
    uint32_t w[32];
    for (int j = 0; j < 32; ++j) w[j] = in[t * 32 + j];
    for (int j = 0; j < 64; ++j) lds[t * 64 + j] = 0;
    __syncthreads();
    if (!skip[t]) lds_append_flagged64(lds_address(lds) + t * 256, flags[t], w);
    lds_append_complete();
    __syncthreads();
    for (int j = 0; j < 64; ++j) out[t * 64 + j] = lds[t * 64 + j];
}
}  // namespace ndzip_hip
"""


@pytest.fixture(scope="module")
def asm_objects(tmp_path_factory):
    from tests import gfx950_exec as gx

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path_factory.mktemp("asmblocks")
    (d / "b.hip").write_text(ASM_SOURCE)
    out = {}
    for name, flags in (("assembled", []), ("compiled", ["-DNDZIP_NO_EXEC_ASM"])):
        co = d / f"{name}.hsaco"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "--no-gpu-bundle-output", "-I", os.path.join(root, "ndzip_amd", "csrc"), *flags,
                            str(d / "b.hip"), "-o", str(co)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = gx.CodeObject(str(co))
    return out


ASM_KERNELS = {  # name: (dynamic LDS bytes, [(argument name, words, "in" | "out")])
    **{f"k_row{d}": (0, [("data", 1024, "in"), ("out", 1024, "out")]) for d in (1, 2, 4, 8)},
    "k_scan64": (0, [("data", 128, "in"), ("out", 128, "out")]),
    "k_pair": (0, [("data", 512, "in"), ("out", 512, "out")]),
    "k_append": (64 * 32 * 4, [("data", 2048, "in"), ("out", 2048, "out"), ("ends", 64, "out"), ("skip", 64, "in")]),
    "k_append64": (64 * 64 * 4, [("data", 2048, "in"), ("flags", 64, "in"), ("out", 4096, "out"), ("skip", 64, "in")]),
}

ASM_HOST_MAIN = r"""
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
// model_blocks <kernel> <dynamic LDS bytes> <file per argument, in order>: inputs are read, outputs (size from the file) written back
int main(int argc, char **argv) {
    const std::string k = argv[1];
    const size_t lds = strtoul(argv[2], nullptr, 10);
    std::vector<std::vector<uint32_t>> a;
    for (int i = 3; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) return 2;
        fseek(f, 0, SEEK_END);
        a.emplace_back(static_cast<size_t>(ftell(f)) / 4);
        fseek(f, 0, SEEK_SET);
        if (fread(a.back().data(), 4, a.back().size(), f) != a.back().size()) return 2;
        fclose(f);
    }
    using namespace ndzip_hip;
    auto p = [&](int i) { return a[i].data(); };
    if (k == "k_row1") hipLaunchKernelGGL(k_row1, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_row2") hipLaunchKernelGGL(k_row2, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_row4") hipLaunchKernelGGL(k_row4, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_row8") hipLaunchKernelGGL(k_row8, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_scan64") hipLaunchKernelGGL(k_scan64, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_pair") hipLaunchKernelGGL(k_pair, dim3(1), dim3(64), lds, nullptr, p(0), p(1));
    else if (k == "k_append") hipLaunchKernelGGL(k_append, dim3(1), dim3(64), lds, nullptr, p(0), p(1), p(2), p(3));
    else if (k == "k_append64") hipLaunchKernelGGL(k_append64, dim3(1), dim3(64), lds, nullptr, p(0), p(1), p(2), p(3));
    else return 3;
    for (int i = 3; i < argc; ++i) {
        FILE *f = fopen(argv[i], "wb");
        if (!f || fwrite(a[i - 3].data(), 4, a[i - 3].size(), f) != a[i - 3].size()) return 4;
        fclose(f);
    }
    return 0;
}
"""


@pytest.fixture(scope="module")
def model_blocks(tmp_path_factory):
    """The same kernels on the functional model: host compiler, tests/wavesim -- and therefore the MODEL's own C++ versions of the five
    blocks (tests/wavesim/gfx950_lds.hpp substitutes the product header there: assembly has no host meaning)."""
    from tests.wavesim import build as simbuild

    here = os.path.dirname(os.path.abspath(simbuild.__file__))
    d = tmp_path_factory.mktemp("modelblocks")
    # (on the model the workgroup's LDS is the array ndzip_hip::smem of its runtime header: the block-scope extern declaration goes)
    model_source = ASM_SOURCE.replace('extern "C" __global__', "__global__").replace("    extern __shared__ __attribute__((aligned(128))) char smem[];\n", "")
    assert "    extern __shared__" not in model_source
    (d / "b.cc").write_text(model_source + ASM_HOST_MAIN)
    exe = d / "model_blocks"
    r = subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", here, str(d / "b.cc"), os.path.join(here, "wavesim.cc"), "-ldl",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(exe), d


def _both(asm_objects, kernel, make_args, model_blocks=None):
    """run `kernel` from both code objects (interpreter) and on the functional model on identical inputs; returns {form: (outputs, opcodes)}"""
    lds, spec = ASM_KERNELS[kernel]
    res = {}
    for form, co in asm_objects.items():
        args, outs = make_args()
        ops = _run(co, kernel, 1, 64, *args, lds=lds)
        res[form] = (outs, ops)
    if model_blocks is not None:
        exe, d = model_blocks
        args, outs = make_args()
        files = []
        for i, arr in enumerate(args):
            f = d / f"{kernel}_{i}.bin"
            arr.tofile(f)
            files.append(str(f))
        r = subprocess.run([exe, kernel, str(lds), *files], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (kernel, r.returncode, r.stdout[-500:], r.stderr[-1500:])
        for i, arr in enumerate(args):   # read everything back: the outputs are among the arguments
            arr[...] = np.fromfile(files[i], dtype=np.uint32)
        res["model"] = (outs, set())
    return res


def _pairs64(words, n):
    """[64 lanes][n] uint64 from the kernels' layout: per lane n low dwords, then n high dwords"""
    w = words.reshape(64, 2 * n).astype(np.uint64)
    return w[:, :n] | (w[:, n:] << np.uint64(32))


@pytest.mark.parametrize("D", [1, 2, 4, 8])
def test_row_scan_step64_both_forms_are_the_row_shifted_sum(asm_objects, D, model_blocks):
    rng = np.random.default_rng(10 + D)
    data = rng.integers(0, 1 << 32, size=64 * 16, dtype=np.uint64).astype(np.uint32)
    data[:16] = 0xFFFFFFFF  # (carries out of the low dword)

    def make():
        out = np.zeros(64 * 16, dtype=np.uint32)
        return (data, out), out

    res = _both(asm_objects, f"k_row{D}", make, model_blocks)
    assert "v_add_co_u32_dpp" in res["assembled"][1] and "v_add_co_u32_dpp" not in res["compiled"][1]
    x = _pairs64(data, 8)
    lane = np.arange(64)
    shifted = np.where(((lane % 16) >= D)[:, None], x[np.maximum(lane - D, 0)], np.uint64(0))   # row_shr:D, zeros shifted in (bound_ctrl)
    want = x + shifted
    for form in res:
        assert np.array_equal(_pairs64(res[form][0], 8), want), form


def test_wave_inclusive_scan64_both_forms_are_the_prefix_sum(asm_objects, model_blocks):
    rng = np.random.default_rng(20)
    data = rng.integers(0, 1 << 32, size=128, dtype=np.uint64).astype(np.uint32)
    data[0:40:2] = 0xFFFFFFFF

    def make():
        out = np.zeros(128, dtype=np.uint32)
        return (data, out), out

    res = _both(asm_objects, "k_scan64", make, model_blocks)
    x = data[0::2].astype(np.uint64) | (data[1::2].astype(np.uint64) << np.uint64(32))
    want = np.cumsum(x, dtype=np.uint64)
    for form in res:
        got = res[form][0][0::2].astype(np.uint64) | (res[form][0][1::2].astype(np.uint64) << np.uint64(32))
        assert np.array_equal(got, want), form


def test_pair_exchange_select4_both_forms(asm_objects, model_blocks):
    rng = np.random.default_rng(30)
    data = rng.integers(0, 1 << 32, size=64 * 8, dtype=np.uint64).astype(np.uint32)

    def make():
        out = np.zeros(64 * 8, dtype=np.uint32)
        return (data, out), out

    res = _both(asm_objects, "k_pair", make, model_blocks)
    assert "v_cndmask_b32_dpp" in res["assembled"][1]
    v = data.reshape(64, 8)
    a, b = v[:, :4], v[:, 4:]
    lane = np.arange(64)
    odd = (lane & 1).astype(bool)[:, None]
    other = lane ^ 1
    want = np.concatenate([np.where(odd, b, a[other]), np.where(odd, b[other], a)], axis=1)  # lo = odd ? own b : the other lane's a; hi = odd ? the other lane's b : own a
    for form in res:
        assert np.array_equal(res[form][0].reshape(64, 8), want), form


def test_lds_append_nonzero_both_forms_under_a_partial_exec_mask(asm_objects, model_blocks):
    rng = np.random.default_rng(40)
    data = rng.integers(0, 1 << 32, size=64 * 32, dtype=np.uint64).astype(np.uint32)
    data[rng.random(64 * 32) < 0.45] = 0
    data[3 * 32:4 * 32] = 0          # a lane that keeps nothing
    data[4 * 32:5 * 32] |= 1         # ... and one that keeps everything

    skip = (np.arange(64) % 5 == 2).astype(np.uint32)

    def make():
        out, ends = np.zeros(64 * 32, dtype=np.uint32), np.zeros(64, dtype=np.uint32)
        return (data, out, ends, skip), (out, ends)

    res = _both(asm_objects, "k_append", make, model_blocks)
    assert "v_cmpx_ne_u32_e32" in res["assembled"][1] and "v_cmpx_ne_u32_e32" not in res["compiled"][1]
    want, ends = np.zeros((64, 32), dtype=np.uint32), np.zeros(64, dtype=np.uint32)
    for t in range(64):
        if t % 5 == 2:
            ends[t] = 0xFFFFFFFF
            continue
        kept = data[t * 32:(t + 1) * 32]
        kept = kept[kept != 0]
        want[t, :len(kept)] = kept
        ends[t] = 4 * len(kept)
    for form in res:
        out, got_ends = res[form][0]
        assert np.array_equal(out.reshape(64, 32), want) and np.array_equal(got_ends, ends), form


def test_lds_append_flagged64_both_forms_under_a_partial_exec_mask(asm_objects, model_blocks):
    rng = np.random.default_rng(50)
    data = rng.integers(0, 1 << 32, size=64 * 32, dtype=np.uint64).astype(np.uint32)
    flags = rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32)
    flags[:4] = [0, 0xFFFFFFFF, 0x80000000, 1]

    skip = (np.arange(64) % 7 == 3).astype(np.uint32)

    def make():
        out = np.zeros(64 * 64, dtype=np.uint32)
        return (data, flags, out, skip), out

    res = _both(asm_objects, "k_append64", make, model_blocks)
    assert "v_cmpx_gt_i32_e32" in res["assembled"][1] and "v_bitop3_b32" in res["assembled"][1]
    want = np.zeros((64, 64), dtype=np.uint32)
    for t in range(64):
        if t % 7 == 3:
            continue
        a = t * 256   # (the swizzle depends on the address within the workgroup's LDS: lds[] starts at 0 in these kernels)
        for i in range(32):
            if (int(flags[t]) >> (31 - i)) & 1:
                at = a ^ ((a >> 3) & 0x70)
                want[at // 256, (at % 256) // 4] = data[t * 32 + i]
                a += 8
    for form in res:
        assert np.array_equal(res[form][0].reshape(64, 64), want), form


# ---- the two emulators against each other on DPP: every control the kernels use (and their neighbours), full and partial EXEC ---------
#
# The same source twice: hipcc -> gfx950 code object -> the interpreter; host compiler + tests/wavesim (the functional model the
# kernels' C++ runs on in most of this suite).  The interpreter's DPP reading is anchored above (LLVM's own wave scan, the compiled
# fallbacks of the asm blocks); the model's must give the same lanes, including what happens to `old` under row / bank masks, bound_ctrl
# and inactive source lanes.

DPP_COMBOS = [(ctrl, rm, bm, bc) for ctrl in (0xB1, 0x4E, 0x1B, 0x111, 0x112, 0x114, 0x118, 0x101, 0x104, 0x121, 0x128, 0x138, 0x130, 0x140, 0x141, 0x142, 0x143)
              for rm, bm, bc in ((0xF, 0xF, 1), (0xF, 0xF, 0), (0xA, 0xF, 0), (0xC, 0xF, 0), (0xF, 0xA, 0), (0x5, 0x6, 1))]

DPP_SOURCE = "#include <hip/hip_runtime.h>\n#include <cstdint>\n" + r"""
template<int Ctrl, int RowMask, int BankMask, bool Bound>
__device__ uint32_t one(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v ^ 0x5a5a5a5au), static_cast<int>(v), Ctrl, RowMask, BankMask, Bound));
}
__global__ void k_dpp(const uint32_t *in, const uint32_t *skip, uint32_t *out) {
    const int t = threadIdx.x;
    const uint32_t v = in[t];
    uint32_t *full = out + t * COUNT * 2, *part = full + COUNT;
    int c = 0;
#define STEP(ctrl, rm, bm, bc) full[c++] = one<ctrl, rm, bm, bc>(v);
    COMBOS
#undef STEP
    for (int k = 0; k < COUNT; ++k) part[k] = 0xeeeeeeeeu;
    if (!skip[t]) {
        c = 0;
#define STEP(ctrl, rm, bm, bc) part[c++] = one<ctrl, rm, bm, bc>(v);
        COMBOS
#undef STEP
    }
}
""".replace("COMBOS", " ".join(f"STEP({c}, {rm}, {bm}, {'true' if bc else 'false'})" for c, rm, bm, bc in DPP_COMBOS)).replace("COUNT", str(len(DPP_COMBOS)))

DPP_HOST_MAIN = r"""
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    std::vector<uint32_t> in(64), skip(64), out(64 * COUNT * 2);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), 4, 64, f) != 64 || fread(skip.data(), 4, 64, f) != 64) return 2;
    fclose(f);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, nullptr, in.data(), skip.data(), out.data());
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) return 3;
    fclose(f);
    return 0;
}
""".replace("COUNT", str(len(DPP_COMBOS)))


def test_the_functional_model_and_the_interpreter_agree_on_dpp(tmp_path):
    from tests import gfx950_exec as gx
    from tests.wavesim import build as simbuild

    here = os.path.dirname(os.path.abspath(simbuild.__file__))
    rng = np.random.default_rng(60)
    v = rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32)
    skip = (rng.random(64) < 0.35).astype(np.uint32)
    skip[[15, 16, 31, 32]] = [1, 0, 1, 1]        # row edges: the lanes a row_shr / row_bcast reads are inactive
    n = len(DPP_COMBOS)
    # the interpreter on hipcc's code
    (tmp_path / "d.hip").write_text(DPP_SOURCE.replace("__global__", 'extern "C" __global__'))
    co = tmp_path / "d.hsaco"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "--no-gpu-bundle-output", str(tmp_path / "d.hip"), "-o", str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out_i = np.zeros(64 * n * 2, dtype=np.uint32)
    ops = _run(gx.CodeObject(str(co)), "k_dpp", 1, 64, v, skip, out_i)
    assert any(o.endswith("_dpp") for o in ops)
    # the model on the host compiler's code
    (tmp_path / "d.cc").write_text(DPP_SOURCE + DPP_HOST_MAIN)
    exe = tmp_path / "dpp_model"
    r = subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", here, str(tmp_path / "d.cc"), os.path.join(here, "wavesim.cc"), "-ldl",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    np.concatenate([v, skip]).tofile(tmp_path / "in.bin")
    r = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    out_m = np.fromfile(tmp_path / "out.bin", dtype=np.uint32)
    a, b = out_i.reshape(64, 2, n), out_m.reshape(64, 2, n)
    bad = np.argwhere(a != b)
    assert bad.size == 0, [(int(t), "partial" if h else "full", tuple(hex(x) for x in DPP_COMBOS[c]), hex(int(a[t, h, c])), hex(int(b[t, h, c]))) for t, h, c in bad[:6]]
    assert (a[skip.astype(bool), 1] == 0xEEEEEEEE).all()   # lanes that sat the partial pass out wrote nothing


LANES_SOURCE = "#include <hip/hip_runtime.h>\n#include <cstdint>\n" + r"""
__global__ void k_lanes2(const uint32_t *in, const uint32_t *skip, uint32_t *out) {
    const int t = threadIdx.x;
    const uint32_t v = in[t];
    uint32_t *o = out + t * 12;
    o[0] = __shfl_xor(v, 1);
    o[1] = __shfl_xor(v, 8);
    o[2] = __shfl_xor(v, 32);
    o[3] = __shfl_up(v, 1u);
    o[4] = __shfl_up(v, 9u);
    o[5] = __shfl(v, 13);
    o[6] = __shfl(v, (t * 11 + 5) & 63);
    o[7] = __shfl_up(v, 3u, 16);
    const unsigned long long m = __ballot((v >> 2) & 1);
    o[8] = (uint32_t) m;
    o[9] = (uint32_t) (m >> 32);
    o[10] = 0xeeeeeeeeu;
    o[11] = 0xeeeeeeeeu;
    if (!skip[t]) {   // a ballot among the lanes that are left: the others' bits are 0 whatever they voted above
        const unsigned long long p = __ballot(v & 1);
        o[10] = (uint32_t) p;
        o[11] = (uint32_t) (p >> 32);
    }
}
"""

LANES_HOST_MAIN = r"""
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    std::vector<uint32_t> in(64), skip(64), out(64 * 12);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), 4, 64, f) != 64 || fread(skip.data(), 4, 64, f) != 64) return 2;
    fclose(f);
    hipLaunchKernelGGL(k_lanes2, dim3(1), dim3(64), 0, nullptr, in.data(), skip.data(), out.data());
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) return 3;
    fclose(f);
    return 0;
}
"""


def test_shuffles_and_ballots_model_interpreter_and_the_definition(tmp_path):
    """The functional model's __shfl / __shfl_up / __shfl_xor / __ballot (tests/wavesim/hip/hip_runtime.h) and the interpreter's execution
    of hipcc's lowering of the same source, both against what HIP defines -- including a ballot under a partial EXEC mask."""
    from tests import gfx950_exec as gx
    from tests.wavesim import build as simbuild

    here = os.path.dirname(os.path.abspath(simbuild.__file__))
    rng = np.random.default_rng(70)
    v = rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32)
    skip = (rng.random(64) < 0.4).astype(np.uint32)
    lane = np.arange(64)
    want = np.zeros((64, 12), dtype=np.uint32)
    want[:, 0], want[:, 1], want[:, 2] = v[lane ^ 1], v[lane ^ 8], v[lane ^ 32]
    want[:, 3] = v[np.where(lane >= 1, lane - 1, lane)]
    want[:, 4] = v[np.where(lane >= 9, lane - 9, lane)]
    want[:, 5] = v[13]
    want[:, 6] = v[(lane * 11 + 5) & 63]
    want[:, 7] = v[np.where((lane & 15) >= 3, lane - 3, lane)]
    m = sum(int((v[k] >> 2) & 1) << k for k in range(64))
    want[:, 8], want[:, 9] = m & 0xFFFFFFFF, m >> 32
    p = sum(int(v[k] & 1) << k for k in range(64) if not skip[k])
    want[:, 10] = np.where(skip, 0xEEEEEEEE, p & 0xFFFFFFFF)
    want[:, 11] = np.where(skip, 0xEEEEEEEE, p >> 32)
    # the interpreter
    (tmp_path / "l.hip").write_text(LANES_SOURCE.replace("__global__", 'extern "C" __global__'))
    co = tmp_path / "l.hsaco"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "--no-gpu-bundle-output", str(tmp_path / "l.hip"), "-o", str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out_i = np.zeros(64 * 12, dtype=np.uint32)
    _run(gx.CodeObject(str(co)), "k_lanes2", 1, 64, v, skip, out_i)
    assert np.array_equal(out_i.reshape(64, 12), want), np.argwhere(out_i.reshape(64, 12) != want)[:6]
    # the model
    (tmp_path / "l.cc").write_text(LANES_SOURCE + LANES_HOST_MAIN)
    exe = tmp_path / "lanes_model"
    r = subprocess.run([simbuild.CXX, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", here, str(tmp_path / "l.cc"), os.path.join(here, "wavesim.cc"), "-ldl",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    np.concatenate([v, skip]).tofile(tmp_path / "in.bin")
    r = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    out_m = np.fromfile(tmp_path / "out.bin", dtype=np.uint32).reshape(64, 12)
    assert np.array_equal(out_m, want), np.argwhere(out_m != want)[:6]


ATOMICS_SOURCE = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
// tickets drawn by one lane per wavefront (the compress kernels' scheme), an exchange chain, a fetch-or, and v_readfirstlane under a
// partial EXEC mask (the first ACTIVE lane's value)
extern "C" __global__ void k_atomics(uint32_t *counter, uint32_t *tickets, uint32_t *word, uint32_t *seen, uint32_t *flags, const uint32_t *in,
        const uint32_t *skip, uint32_t *first) {
    const int t = threadIdx.x, wave = blockIdx.x * (blockDim.x / 64) + t / 64;
    if ((t & 63) == 0) {
        tickets[wave] = atomicAdd(counter, 1u);                       // returning atomic: every wavefront a distinct ticket
        seen[wave] = atomicExch(word, 1000u + wave);                   // the chain of previous values ends in the initial one
        atomicOr(flags, 1u << (wave & 31));
    }
    const uint32_t v = in[blockIdx.x * blockDim.x + t];
    first[blockIdx.x * blockDim.x + t] = 0xeeeeeeeeu;
    if (!skip[t & 63]) first[blockIdx.x * blockDim.x + t] = __builtin_amdgcn_readfirstlane(v);
}
"""


def test_returning_atomics_and_readfirstlane_under_a_partial_exec_mask(tmp_path):
    from tests import gfx950_exec as gx

    (tmp_path / "a.hip").write_text(ATOMICS_SOURCE)
    co = tmp_path / "a.hsaco"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--genco", "--no-gpu-bundle-output", str(tmp_path / "a.hip"), "-o", str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    grid, block = 5, 128
    waves = grid * block // 64
    rng = np.random.default_rng(80)
    v = rng.integers(0, 1 << 32, size=grid * block, dtype=np.uint64).astype(np.uint32)
    skip = (rng.random(64) < 0.5).astype(np.uint32)
    skip[:3] = 1                                     # lanes 0-2 are inactive: the first active lane is not lane 0
    counter, word, flags = np.array([7], dtype=np.uint32), np.array([42], dtype=np.uint32), np.array([0], dtype=np.uint32)
    tickets, seen, first = np.zeros(waves, dtype=np.uint32), np.zeros(waves, dtype=np.uint32), np.zeros(grid * block, dtype=np.uint32)
    ops = _run(gx.CodeObject(str(co)), "k_atomics", grid, block, counter, tickets, word, seen, flags, v, skip, first)
    assert "v_readfirstlane_b32" in ops and any(o.startswith("global_atomic_add") for o in ops) and any(o.startswith("global_atomic_swap") for o in ops)
    assert sorted(tickets.tolist()) == list(range(7, 7 + waves)) and int(counter[0]) == 7 + waves
    # the exchange chain: every wavefront saw the initial value or another wavefront's, each exactly once, and the last writer's is left
    values = [42] + [1000 + w for w in range(waves)]
    assert sorted(seen.tolist() + [int(word[0])]) == sorted(values)
    assert int(flags[0]) == sum(1 << (w & 31) for w in set(w & 31 for w in range(waves)))
    lead = int(np.flatnonzero(skip == 0)[0])
    for w0 in range(0, grid * block, 64):
        want = np.where(skip.astype(bool), np.uint32(0xEEEEEEEE), v[w0 + lead])
        assert np.array_equal(first[w0:w0 + 64], want)


MEMORY_SOURCE = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
struct u3 { uint32_t a, b, c; };
// data movement in every width and pairing the compiler likes to use: 4 / 8 / 12 / 16-byte global accesses, LDS reads and writes that it
// fuses into ds_read2_b32 / ds_read2_b64 / ds_read2st64_b32 / ds_write2_* when two accesses sit a constant stride apart
extern "C" __global__ void k_memory(const uint32_t *in, uint32_t *out) {
    __shared__ uint32_t a[2048];
    __shared__ uint2 b[512];
    const int t = threadIdx.x;                                   // 128 work-items
    const uint4 q = reinterpret_cast<const uint4 *>(in)[t];      // words 0 .. 511
    const uint2 d = reinterpret_cast<const uint2 *>(in + 512)[t];  // 512 .. 767
    const u3 e = reinterpret_cast<const u3 *>(in + 768)[t];      // 768 .. 1151
    a[t] = q.x; a[t + 64 * 4] = q.y;                             // a pair 256 words apart: ds_write2st64_b32
    a[1024 + 2 * t] = q.z; a[1025 + 2 * t] = q.w ^ 1u;           // neighbours: ds_write2_b32 / ds_write_b64
    b[t] = d; b[t + 128] = make_uint2(e.a, e.b);                 // 8-byte stores 1 KiB apart: ds_write2st64_b64
    a[1536 + t] = e.c;
    __syncthreads();
    const uint32_t r0 = a[(t * 5) & 127] + a[((t * 5) & 127) + 256];     // ds_read2st64_b32
    const uint2 r1 = b[127 - t], r2 = b[255 - t];                         // ds_read2st64_b64 or two b64
    const uint32_t r3 = a[1536 + ((t + 1) & 127)], r4 = a[1536 + ((t + 2) & 127)];
    uint4 o;
    o.x = r0; o.y = r1.x ^ r2.y; o.z = r1.y + r2.x; o.w = r3 - r4;
    reinterpret_cast<uint4 *>(out)[t] = o;
    u3 p{a[1024 + 2 * ((t + 3) & 127)], a[1025 + 2 * ((t + 7) & 127)], e.c};
    reinterpret_cast<u3 *>(out + 512)[t] = p;
}
"""


def test_memory_operations_in_every_width_the_compiler_pairs(tmp_path):
    from tests import gfx950_exec as gx

    (tmp_path / "m.hip").write_text(MEMORY_SOURCE)
    co = tmp_path / "m.hsaco"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--genco", "--no-gpu-bundle-output", str(tmp_path / "m.hip"), "-o", str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(90)
    x = rng.integers(0, 1 << 32, size=1152, dtype=np.uint64).astype(np.uint32)
    out = np.zeros(512 + 384, dtype=np.uint32)
    ops = _run(gx.CodeObject(str(co)), "k_memory", 1, 128, x, out)
    wide = {o for o in ops if o.startswith(("ds_read2", "ds_write2", "global_load_dwordx", "global_store_dwordx", "ds_read_b64", "ds_write_b64", "ds_read_b128"))}
    assert any(o.startswith("ds_read2") or o.startswith("ds_write2") for o in wide) and "global_load_dwordx4" in wide and "global_load_dwordx3" in wide, sorted(wide)
    t = np.arange(128)
    q, d, e = x[:512].reshape(128, 4), x[512:768].reshape(128, 2), x[768:].reshape(128, 3)
    a = np.zeros(2048, dtype=np.uint32)
    a[t], a[t + 256] = q[:, 0], q[:, 1]
    a[1024 + 2 * t], a[1025 + 2 * t] = q[:, 2], q[:, 3] ^ np.uint32(1)
    b = np.zeros((512, 2), dtype=np.uint32)
    b[t], b[t + 128] = d, e[:, :2]
    a[1536 + t] = e[:, 2]
    r0 = a[(t * 5) & 127] + a[((t * 5) & 127) + 256]
    r1, r2 = b[127 - t], b[255 - t]
    r3, r4 = a[1536 + ((t + 1) & 127)], a[1536 + ((t + 2) & 127)]
    want = np.stack([r0, r1[:, 0] ^ r2[:, 1], r1[:, 1] + r2[:, 0], r3 - r4], axis=1).astype(np.uint32)
    assert np.array_equal(out[:512].reshape(128, 4), want)
    want3 = np.stack([a[1024 + 2 * ((t + 3) & 127)], a[1025 + 2 * ((t + 7) & 127)], e[:, 2]], axis=1)
    assert np.array_equal(out[512:].reshape(128, 3), want3)
