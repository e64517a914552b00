"""An independent reading of the DPP wave scan (no GPU, no functional model): LLVM's own AMDGPU atomic optimizer builds a wave64
inclusive scan for gfx9-family targets when it pre-reduces a divergent atomicAdd (strategy DPP) -- four row shifts, row_bcast:15
into rows 1 and 3, row_bcast:31 into rows 2 and 3.  The product's wave_inclusive_scan (ndzip_amd/csrc/codec_kernels.hpp) must
compile to the very same sequence of DPP controls, row masks, bank masks and bound_ctrl on gfx950: then its lane semantics are the
compiler writers' reading of the ISA, not only this repo's (tests/wavesim models them from the ISA manual; the GPU suite checks
them on silicon in test_hip_stages.py::test_wave_scan_and_sum)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

REFERENCE = r"""
#include <hip/hip_runtime.h>
__global__ void llvm_scan(int *p, const int *in, int *out) { out[threadIdx.x] = atomicAdd(p, in[threadIdx.x]); }
"""

PRODUCT = r"""
#include "codec_kernels.hpp"
__global__ void product_scan(const uint32_t *in, uint32_t *out) {
    out[threadIdx.x] = ndzip_hip::wave_inclusive_scan(in[threadIdx.x], static_cast<int>(threadIdx.x & 63u));
}
"""


def _dpp_sequence(asm, kernel):
    body = re.search(rf"^{kernel}:.*?s_endpgm", asm, re.S | re.M)
    if body is None:  # (mangled name)
        body = re.search(rf"^_Z\w*{kernel}\w*:.*?s_endpgm", asm, re.S | re.M)
    assert body, kernel
    seq = []
    for line in body.group(0).splitlines():
        m = re.search(r"v_add_u32_dpp\s+\S+,\s*\S+,\s*\S+\s+(.*)$", line.split(";")[0].rstrip())
        if m:
            seq.append(" ".join(m.group(1).split()))
    return seq


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_wave_scan_is_the_sequence_llvm_generates_for_its_own_wave64_scan(tmp_path):
    out = {}
    for name, src, flags in (("llvm", REFERENCE, ["-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"]),
                             ("product", PRODUCT, ["-I", os.path.join(ROOT, "ndzip_amd", "csrc")])):
        f = tmp_path / f"{name}.hip"
        f.write_text(src)
        s = tmp_path / f"{name}.s"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags, str(f), "-o", str(s)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = _dpp_sequence(s.read_text(), "llvm_scan" if name == "llvm" else "product_scan")
    want = ["row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]
    assert out["llvm"] == want, out["llvm"]        # what LLVM emits for its own scan on gfx950 (wave64)
    assert out["product"] == out["llvm"], out["product"]
