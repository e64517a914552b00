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


def _run(co, kernel, grid, block, *args):
    from tests import gfx950_exec as gx

    k = gx.Kernel(co, kernel)
    assert not k.missing, f"{kernel}: the compiler used {k.missing}, which the interpreter does not know (change the SOURCE, not the interpreter)"
    packed = b"".join(struct.pack("<Q", a.ctypes.data) if isinstance(a, np.ndarray) else struct.pack("<i", a) for a in args)
    gx.run_grid(k, grid, block, 0, packed, resident=4, quantum=500)
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
