"""Build the gfx950 HIP libraries in-tree: ndzip_amd/libndzip_hip.so (the product: include/ndzip_hip.h) and
ndzip_amd/libndzip_hip_stages.so (parity-test hooks only: include/ndzip_hip_stages.h -- the single-hypercube stage kernels, which
the product library does not contain).

`python -m ndzip_amd.build` (or `__graft_entry__.build()`) cross-compiles without a GPU.  The three
translation units are compiled in parallel; objects go to ndzip_amd/csrc/_build/ (git-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libndzip_hip.so")
STAGES_OUT = os.path.join(HERE, "libndzip_hip_stages.so")
OBJDIR = os.path.join(CSRC, "_build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
SOURCES = ["kernels_f32.hip", "kernels_f64.hip", "capi.hip"]
STAGE_SOURCES = ["stages_f32.hip", "stages_f64.hip", "stages_capi.hip"]  # -> libndzip_hip_stages.so (test hooks)
# every header a translation unit can see: a stale object for the newest kernel is the worst kind of benchmark bug
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inl"))) + ["../../include/ndzip_hip.h", "../../include/ndzip_hip_stages.h"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# No atomic optimizer: it rewrites the single-lane ticket atomicAdd into mbcnt + atomic + readfirstlane and waits for the
# result on the spot, which puts the atomic's round trip back on the path the kernel takes care to hide it behind the copy-out.
FLAGS += ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
FLAGS += os.environ.get("NDZIP_EXTRA_FLAGS", "").split()  # experiments only (tools/)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs, stage_objs = [], []
    for src in SOURCES + STAGE_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        (objs if src in SOURCES else stage_objs).append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if "-c" in cmd:  # keep the per-kernel register / scratch / occupancy remarks next to the object (kernel_resources())
            with open(cmd[cmd.index("-o") + 1] + ".resources.txt", "w") as f:
                f.write(r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=len(SOURCES) + len(STAGE_SOURCES)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT, *objs])
    if force or jobs or _stale(STAGES_OUT, stage_objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", STAGES_OUT, *stage_objs])
    build_cli(force=force, verbose=verbose)
    build_rccl(force=force, verbose=verbose)
    return OUT


def kernels_fingerprint() -> str:
    """Short hash over the device-code sources (csrc/*.hip, *.hpp, *.inl) and the compiler flags: what a set of hardware
    counters was measured on.  profiles/traffic.json entries carry it, and bench.py reports `roofline.traffic` only for an
    entry whose fingerprint is the one of the sources in this tree (counters of other kernels next to a fresh launch time would
    be a made-up number)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".inl")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_resources() -> dict:
    """{demangled-ish kernel name: {"vgprs", "scratch", "occupancy", "sgpr_spill"}} of the last build of every translation unit
    (hipcc -Rpass-analysis=kernel-resource-usage).  A compress kernel with scratch is a performance bug: a scratch reload is a
    vector-memory load and waits for every prefetch load issued before it."""
    import re

    out = {}
    for src in SOURCES:
        path = os.path.join(OBJDIR, src.replace(".hip", ".o")) + ".resources.txt"
        if not os.path.exists(path):
            continue
        name = None
        for line in open(path):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {}
                continue
            if name is None:
                continue
            for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)")):
                m = re.search(pat, line)
                if m:
                    out[name][key] = int(m.group(1))
    return out


CLI_TOOLS = {  # binary next to the library it loads: source
    os.path.join(HERE, "ndzip-hip"): os.path.join(HERE, "cli", "ndzip_hip_cli.cc"),
    os.path.join(HERE, "ndzip-hip-benchmark"): os.path.join(HERE, "cli", "ndzip_hip_benchmark.cc"),
}
CLI_OUT = os.path.join(HERE, "ndzip-hip")
BENCHMARK_OUT = os.path.join(HERE, "ndzip-hip-benchmark")


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """The file-level tools (plain C++ over the C ABI, no HIP headers): ndzip_amd/ndzip-hip (the reference's `compress`) and
    ndzip_amd/ndzip-hip-benchmark (the `ndzip-hip` rows of the reference's benchmark CSV)."""
    for out, src in CLI_TOOLS.items():
        if force or _stale(out, [src, OUT, os.path.join(HERE, "..", "include", "ndzip_hip.h")]):
            cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-o", out, src, "-L" + HERE, "-lndzip_hip", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"g++ failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return CLI_OUT


RCCL_OUT = os.path.join(HERE, "libndzip_hip_rccl.so")
RCCL_SOURCES = ["sharded.cc", "sharded_rccl.cc"]
SHARDED_CLI_OUT = os.path.join(HERE, "ndzip-hip-sharded")
SHARDED_CLI_SRC = os.path.join(HERE, "cli", "ndzip_hip_sharded_cli.cc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_rccl(force: bool = False, verbose: bool = False) -> str:
    """ndzip_amd/libndzip_hip_rccl.so (include/ndzip_hip_sharded.h): the C++ host of the multi-GPU path -- plain host code over the C
    ABI of libndzip_hip.so, the HIP runtime's memory calls and RCCL.  No device code: kernels_fingerprint() does not cover it."""
    srcs = [os.path.join(CSRC, f) for f in RCCL_SOURCES]
    deps = srcs + [OUT, os.path.join(HERE, "..", "include", "ndzip_hip.h"), os.path.join(HERE, "..", "include", "ndzip_hip_sharded.h"), os.path.abspath(__file__)]
    if force or _stale(RCCL_OUT, deps):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra", "-DNDZIP_HIP_BUILD", "-D__HIP_PLATFORM_AMD__",
               "-I" + os.path.join(ROCM, "include"), "-o", RCCL_OUT, *srcs, "-L" + HERE, "-lndzip_hip", "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-Wl,--no-undefined"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    # ndzip_amd/ndzip-hip-sharded: the file-level tool of the multi-GPU path (plain C++ over include/ndzip_hip_sharded.h, no HIP header)
    if force or _stale(SHARDED_CLI_OUT, [SHARDED_CLI_SRC, RCCL_OUT, os.path.join(HERE, "..", "include", "ndzip_hip_sharded.h")]):
        cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", "-o", SHARDED_CLI_OUT, SHARDED_CLI_SRC, "-L" + HERE, "-lndzip_hip_rccl", "-lndzip_hip",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return RCCL_OUT


TEST_VARIANT_SPIN0 = os.path.join(HERE, "_variants", "spin0.so")
TEST_VARIANT_PLAIN = os.path.join(HERE, "_variants", "plain.so")
TEST_VARIANTS = {  # name: extra defines
    "spin0": ["-DNDZIP_LOOKBACK_SPIN_LIMIT=0"],
    "plain": ["-DNDZIP_NO_EXEC_ASM", "-DNDZIP_NO_SCALAR_PINS"],
}


def build_test_variants(force: bool = False, verbose: bool = False) -> str:
    """Test infrastructure, never loaded by the package:
    ndzip_amd/_variants/spin0.so -- the same library with a look-back spin limit of 0 (any wait for a predecessor is a timeout), for
    the GPU test of the give-up path (tests/test_hip_stress.py);
    ndzip_amd/_variants/plain.so -- without the hand-written EXEC-masked assembly and without the v_readfirstlane uniformity pins
    (gfx950_lds.hpp: bisecting aids), for tools/variant_parity.py."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".inl"))] + [os.path.abspath(__file__)]
    jobs, links = [], []
    for name, defines in TEST_VARIANTS.items():
        out = os.path.join(HERE, "_variants", name + ".so")
        objdir = os.path.join(HERE, "_variants", "obj_" + name)
        os.makedirs(objdir, exist_ok=True)
        if not force and not _stale(out, deps):
            continue
        objs = []
        for src in SOURCES:
            o = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(o)
            jobs.append([HIPCC, *FLAGS, *defines, "-c", os.path.join(CSRC, src), "-o", o])
        links.append([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out, *objs])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    for cmd in links:
        run(cmd)
    return TEST_VARIANT_SPIN0


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
