"""TEST INFRASTRUCTURE ONLY: loads the wave64 functional model of the HIP library (tests/wavesim/build.py) and drives it
through the very same C ABI bindings as the real library (ndzip_amd.hip._bind); "device pointers" are numpy buffers."""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import numpy as np

from ndzip_amd import hip

from . import build as simbuild

_sim = {}


def load(variant: str = "", defines=()):
    """WAVESIM_VARIANT=asan in the environment (with the AddressSanitizer runtime LD_PRELOADed, tests/test_wavesim_asan.py)
    makes the default library the sanitised one."""
    if not variant and os.environ.get("WAVESIM_VARIANT") in ("asan", "ubsan"):
        variant = os.environ["WAVESIM_VARIANT"]  # (ubsan: tools/ubsan_rehearsal.sh, with the UBSan runtime LD_PRELOADed)
    if variant not in _sim:
        flags = simbuild.ASAN_FLAGS if variant == "asan" else simbuild.UBSAN_FLAGS if variant == "ubsan" else ()
        kflags = simbuild.LDSPROF_FLAGS if variant == "ldsprof" else ()
        if variant == "ldsprof":
            defines = tuple(defines) + ("WAVESIM_LDSPROF",)
        # (one model library holds the product entry points and the stage hooks: bound for both)
        _sim[variant] = hip._bind_stages(hip._bind(C.CDLL(simbuild.build(variant=variant, defines=defines, extra_flags=flags, kernel_flags=kflags))))
    return _sim[variant]


@contextlib.contextmanager
def active(cus: int = 2, blocks_per_cu: int = 2, variant: str = "", schedule: str = ""):
    """Route ndzip_amd.hip's ctypes calls to the model for the duration of the block (tests only: the product never does).
    schedule: order in which a workgroup's runnable work-items are resumed: "" (forward), "reverse", "random:<seed>"."""
    L = load(variant)
    saved, saved_stages, env = hip._lib, hip._stages_lib, {k: os.environ.get(k) for k in ("WAVESIM_CUS", "WAVESIM_BLOCKS_PER_CU", "WAVESIM_SCHEDULE")}
    os.environ["WAVESIM_CUS"] = str(cus)
    os.environ["WAVESIM_BLOCKS_PER_CU"] = str(blocks_per_cu)
    os.environ["WAVESIM_SCHEDULE"] = schedule
    hip._lib = hip._stages_lib = L
    try:
        yield L
    finally:
        hip._lib, hip._stages_lib = saved, saved_stages
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _words(dtype):
    return np.uint32 if np.dtype(dtype).itemsize == 4 else np.uint64


def compress(data: np.ndarray, cus: int = 2, blocks_per_cu: int = 2, misalign_words: int = 0, schedule: str = "") -> np.ndarray:
    """Device-pointer compress (ndzip_hip_compressor_compress) on the model; returns the stream words."""
    data = np.ascontiguousarray(data)
    with active(cus, blocks_per_cu, schedule=schedule):
        bound = hip.compressed_length_bound(data.dtype, data.shape)
        out = np.zeros(max(1, bound) + 8, dtype=_words(data.dtype))
        length = np.zeros(1, dtype=np.uint32)
        comp = hip.make_hip_compressor(data.dtype, hip.CompressorRequirements(data.shape))
        try:
            comp.compress(data.ctypes.data, data.shape, out.ctypes.data + misalign_words * out.itemsize, length.ctypes.data)
            comp.check()
        finally:
            comp.close()
    n = int(length[0])
    assert n <= bound
    return out[misalign_words:misalign_words + n].copy()


def decompress(stream: np.ndarray, dtype, extent, bounded: bool = False, schedule: str = "", f64_work_items: int = 0) -> np.ndarray:
    stream = np.ascontiguousarray(stream)
    out = np.zeros(extent, dtype=dtype)
    with active(schedule=schedule):
        dec = hip.make_hip_decompressor(dtype, len(extent))
        try:
            buf = stream if stream.size else np.zeros(1, dtype=_words(dtype))
            if f64_work_items:
                dec.set_f64_work_items(f64_work_items)
            dec.decompress(buf.ctypes.data, out.ctypes.data, extent, stream.size if bounded else None)
            dec.check()
            if np.dtype(dtype) == np.float64 and not f64_work_items:
                # (both 64-bit decoder kernels, whatever the library's default is: see tests/util.py::device_decompress)
                for work_items in (128, 256):
                    other = np.full(extent, np.nan, dtype=dtype)
                    dec.set_f64_work_items(work_items)
                    dec.decompress(buf.ctypes.data, other.ctypes.data, extent, stream.size if bounded else None)
                    dec.check()
                    assert np.array_equal(other.view(np.uint64), out.view(np.uint64)), f"the {work_items}-work-item 64-bit decoder differs"
        finally:
            dec.close()
    return out
