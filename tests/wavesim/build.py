"""TEST INFRASTRUCTURE ONLY: builds tests/wavesim/libndzip_hip_wavesim.so -- the product's kernel and C-ABI sources
(ndzip_amd/csrc/*.hip, *.inl, *.hpp, unchanged) compiled as host C++ against the wave64 functional model in this directory
(hip/hip_runtime.h, wavesim.cc).  The one product header that is substituted is gfx950_lds.hpp (a VGPR-pinned LDS address).

Nothing under ndzip_amd/ imports or loads this; tests/test_wavesim_*.py do.  `python -m tests.wavesim.build`."""
from __future__ import annotations

import fcntl
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ndzip_amd", "csrc")
OUT = os.path.join(HERE, "libndzip_hip_wavesim.so")
BUILD = os.path.join(HERE, "_build")
CXX = os.environ.get("WAVESIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-std=c++17", "-O1", "-g0", "-fPIC", "-pthread", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-unused-const-variable"]
FLAGS += os.environ.get("WAVESIM_EXTRA_FLAGS", "").split()
UNITS = ["kernels_f32.hip", "kernels_f64.hip", "capi.hip", "stages_f32.hip", "stages_f64.hip", "stages_capi.hip"]  # (product + stage hooks in ONE model library)


def _sources():
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".inl"))]
    files += [os.path.join(ROOT, "include", "ndzip_hip.h"), os.path.join(ROOT, "include", "ndzip_hip_stages.h")]
    files += [os.path.join(HERE, f) for f in ("gfx950_lds.hpp", "wavesim.cc", os.path.join("hip", "hip_runtime.h"), "build.py")]
    return files


ASAN_FLAGS = ("-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g1")
# undefined behaviour in the kernels' C++ (shift counts, signed overflow, misaligned or out-of-range accesses the host compiler may
# treat differently from hipcc): every finding aborts.  (vptr / function need RTTI and do not apply to this code.)
UBSAN_FLAGS = ("-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-fno-sanitize=vptr,function", "-shared-libsan", "-fno-omit-frame-pointer", "-g1")


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of CXX: what a Python process has to LD_PRELOAD to load the `asan` variant."""
    return subprocess.run([CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


# the LDS access profile (wavesim.cc): every load / store of the product's translation units calls a hook; the scheduler itself is
# compiled without (its hooks must not call themselves)
LDSPROF_FLAGS = ("-fsanitize-coverage=edge,trace-loads,trace-stores", "-g1")  # ("edge": without a coverage type clang instruments nothing)


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=(), extra_flags=(), kernel_flags=()) -> str:
    """variant / defines / extra_flags: a second library built with extra -D or compiler flags (e.g. a tiny look-back spin
    limit, or ASAN_FLAGS for the memory-safety run of tests/test_wavesim_asan.py), suffixed _<variant>."""
    OUT = os.path.join(HERE, f"libndzip_hip_wavesim{'_' + variant if variant else ''}.so")
    BUILD = os.path.join(HERE, "_build", variant or "default")
    FLAGS = list(globals()["FLAGS"]) + [f"-D{d}" for d in defines] + list(extra_flags)
    def fresh():
        return os.path.exists(OUT) and all(os.path.getmtime(f) <= os.path.getmtime(OUT) for f in _sources())

    if not force and fresh():
        return OUT
    # One builder per variant at a time, across processes (the ranks of a torch.distributed.run rehearsal and pytest-xdist workers
    # all come here when a kernel source is newer than the library): the others wait for the lock and find the library fresh.
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", f".lock_{variant or 'default'}"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        return _build_locked(OUT, BUILD, FLAGS, verbose, extra_flags, kernel_flags)


def _build_locked(OUT, BUILD, FLAGS, verbose, extra_flags, kernel_flags) -> str:
    # mirror the product tree so that its relative includes resolve, with the one substituted header
    src = os.path.join(BUILD, "ndzip_amd", "csrc")
    shutil.rmtree(BUILD, ignore_errors=True)
    os.makedirs(src)
    os.makedirs(os.path.join(BUILD, "include"))
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp", ".inl")):
            shutil.copy(os.path.join(CSRC, f), os.path.join(src, f))
    shutil.copy(os.path.join(HERE, "gfx950_lds.hpp"), os.path.join(src, "gfx950_lds.hpp"))
    for h in ("ndzip_hip.h", "ndzip_hip_stages.h"):
        shutil.copy(os.path.join(ROOT, "include", h), os.path.join(BUILD, "include", h))

    def compile_one(job):
        source, obj = job
        cmd = [CXX, *FLAGS, *(kernel_flags if source.endswith(".hip") else ()), "-x", "c++", "-I", HERE, "-c", source, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"wavesim build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return obj

    jobs = [(os.path.join(src, u), os.path.join(BUILD, u.replace(".hip", ".o"))) for u in UNITS]
    jobs.append((os.path.join(HERE, "wavesim.cc"), os.path.join(BUILD, "wavesim.o")))
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(compile_one, jobs))
    tmp = OUT + f".tmp{os.getpid()}"  # (linked next to the library, then renamed: a process that has the old one mapped keeps it)
    cmd = [CXX, "-shared", "-fPIC", "-pthread", *[f for f in extra_flags if f.startswith(("-fsanitize", "-shared-lib"))], "-o", tmp, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"wavesim link failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, OUT)
    return OUT


def build_sharded(force: bool = False, verbose: bool = False, variant: str = None) -> str:
    """tests/wavesim/libndzip_hip_sharded_wavesim.so: the product's multi-GPU host (ndzip_amd/csrc/sharded.cc, unchanged; the RCCL file
    is left out -- the CPU tests supply the collectives table) compiled against the model's runtime header and linked against the model
    library, so that include/ndzip_hip_sharded.h can be driven where there is no GPU.  variant "asan" / "ubsan" (default: what
    WAVESIM_VARIANT says, like sim.load): sharded.cc itself sanitised, linked against the sanitised model."""
    if variant is None:
        variant = os.environ.get("WAVESIM_VARIANT", "") if os.environ.get("WAVESIM_VARIANT") in ("asan", "ubsan") else ""
    extra = ASAN_FLAGS if variant == "asan" else UBSAN_FLAGS if variant == "ubsan" else ()
    model = build(variant=variant, extra_flags=extra)
    out = os.path.join(HERE, f"libndzip_hip_sharded_wavesim{'_' + variant if variant else ''}.so")
    src = os.path.join(CSRC, "sharded.cc")
    deps = [src, model, os.path.join(ROOT, "include", "ndzip_hip.h"), os.path.join(ROOT, "include", "ndzip_hip_sharded.h"), os.path.join(HERE, "hip", "hip_runtime.h"),
            os.path.abspath(__file__)]

    def fresh():
        return os.path.exists(out) and all(os.path.getmtime(f) <= os.path.getmtime(out) for f in deps)

    if not force and fresh():
        return out
    with open(os.path.join(HERE, "_build", f".lock_sharded_{variant or 'default'}"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return out
        tmp = out + f".tmp{os.getpid()}"
        cmd = [CXX, *FLAGS, *extra, "-DNDZIP_HIP_BUILD", "-shared", "-I", HERE, "-o", tmp, src, "-L" + HERE, "-l:" + os.path.basename(model), "-Wl,-rpath,$ORIGIN"]
        if not extra:
            cmd.append("-Wl,--no-undefined")  # (a sanitised library leaves its runtime's symbols to the preloaded runtime)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"wavesim sharded build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
