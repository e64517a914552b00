"""ctypes doorway to the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
module; nothing under ``ndzip_amd/`` does (the product path fails loudly without its HIP library).

Two libraries:
  * ``libndzip_oracle.so``  -- plain-C restatement of the reference algorithm (oracle/ndzip_oracle.c).
  * ``_ref/libndzip_ref.so`` -- the real reference serial CPU path compiled from /root/reference by
    oracle/Makefile (absent => ``have_ref()`` is False; never required at run time on the GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libndzip_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libndzip_ref.so")

_oracle = None
_ref = None

_U32P = C.POINTER(C.c_uint32)


def build(ref: bool = True) -> None:
    """Compile the restatement (always) and the reference (only where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _ext(extent):
    e = (C.c_uint32 * 3)(*([int(x) for x in extent] + [1] * (3 - len(extent))))
    return e


def _bits(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype in (np.dtype(np.float32), np.dtype(np.uint32)):
        return 32
    if dtype in (np.dtype(np.float64), np.dtype(np.uint64)):
        return 64
    raise TypeError(f"unsupported dtype {dtype}")


def lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        L = C.CDLL(_ORACLE_SO)
        for suf in ("u32", "u64"):
            getattr(L, f"ndzip_oracle_compress_{suf}").restype = C.c_uint64
            getattr(L, f"ndzip_oracle_compress_{suf}").argtypes = [C.c_int, _U32P, C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, f"ndzip_oracle_decompress_{suf}").restype = C.c_uint64
            getattr(L, f"ndzip_oracle_decompress_{suf}").argtypes = [C.c_int, _U32P, C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, f"ndzip_oracle_forward_transform_{suf}").restype = None
            getattr(L, f"ndzip_oracle_forward_transform_{suf}").argtypes = [C.c_void_p, C.c_int]
            getattr(L, f"ndzip_oracle_inverse_transform_{suf}").restype = None
            getattr(L, f"ndzip_oracle_inverse_transform_{suf}").argtypes = [C.c_void_p, C.c_int]
            getattr(L, f"ndzip_oracle_transpose_bits_{suf}").restype = None
            getattr(L, f"ndzip_oracle_transpose_bits_{suf}").argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, f"ndzip_oracle_encode_cube_{suf}").restype = C.c_uint32
            getattr(L, f"ndzip_oracle_encode_cube_{suf}").argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, f"ndzip_oracle_decode_cube_{suf}").restype = C.c_uint32
            getattr(L, f"ndzip_oracle_decode_cube_{suf}").argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, f"ndzip_oracle_load_cube_{suf}").restype = None
            getattr(L, f"ndzip_oracle_load_cube_{suf}").argtypes = [C.c_int, _U32P, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ndzip_oracle_num_hypercubes.restype = C.c_uint32
        L.ndzip_oracle_num_hypercubes.argtypes = [C.c_int, _U32P]
        L.ndzip_oracle_border_count.restype = C.c_uint64
        L.ndzip_oracle_border_count.argtypes = [C.c_int, _U32P]
        L.ndzip_oracle_compressed_length_bound.restype = C.c_uint64
        L.ndzip_oracle_compressed_length_bound.argtypes = [C.c_int, C.c_int, _U32P]
        L.ndzip_oracle_border_slices.restype = C.c_uint32
        L.ndzip_oracle_border_slices.argtypes = [C.c_int, _U32P, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32]
        L.ndzip_oracle_max_threads.restype = C.c_int
        _oracle = L
    return _oracle


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(_REF_SO)
        for suf in ("f32", "f64"):
            getattr(L, f"ndzip_ref_compress_{suf}").restype = C.c_int64
            getattr(L, f"ndzip_ref_compress_{suf}").argtypes = [C.c_int, _U32P, C.c_void_p, C.c_void_p]
            getattr(L, f"ndzip_ref_decompress_{suf}").restype = C.c_int64
            getattr(L, f"ndzip_ref_decompress_{suf}").argtypes = [C.c_int, _U32P, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ndzip_ref_compressed_length_bound.restype = C.c_uint64
        L.ndzip_ref_compressed_length_bound.argtypes = [C.c_int, C.c_int, _U32P]
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------------------
# restatement (port)
# ---------------------------------------------------------------------------------------------------------

def num_hypercubes(extent) -> int:
    return int(lib().ndzip_oracle_num_hypercubes(len(extent), _ext(extent)))


def border_count(extent) -> int:
    return int(lib().ndzip_oracle_border_count(len(extent), _ext(extent)))


def compressed_length_bound(dtype, extent) -> int:
    return int(lib().ndzip_oracle_compressed_length_bound(_bits(dtype), len(extent), _ext(extent)))


def border_slices(extent, side: int = 0):
    cap = 1 << 16
    buf = (C.c_uint64 * (2 * cap))()
    n = lib().ndzip_oracle_border_slices(len(extent), _ext(extent), side, buf, cap)
    assert n <= cap
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def compress(data: np.ndarray, num_threads: int = 1, out: np.ndarray = None) -> np.ndarray:
    """Oracle compress of a C-contiguous float32/float64 array -> stream as uint32/uint64 words.
    `out`: optional preallocated (already touched) buffer of compressed_length_bound words -- timing runs reuse it so
    first-touch page faults of a fresh buffer are not billed to the codec; the result is then a view into it."""
    data = np.ascontiguousarray(data)
    bits = _bits(data.dtype)
    wdt = np.uint32 if bits == 32 else np.uint64
    extent = data.shape
    keep = out is not None
    if out is None:
        out = np.zeros(max(1, compressed_length_bound(data.dtype, extent)), dtype=wdt)
    assert out.dtype == wdt and out.size >= max(1, compressed_length_bound(data.dtype, extent))
    fn = getattr(lib(), f"ndzip_oracle_compress_u{bits}")
    n = fn(len(extent), _ext(extent), data.ctypes.data, out.ctypes.data, int(num_threads))
    return out[: int(n)] if keep else out[: int(n)].copy()


def decompress(stream: np.ndarray, dtype, extent, num_threads: int = 1, out: np.ndarray = None):
    """Oracle decompress -> (array of `dtype` with shape `extent`, words consumed).  `out`: optional reusable buffer."""
    bits = _bits(dtype)
    stream = np.ascontiguousarray(stream)
    assert _bits(stream.dtype) == bits
    if out is None:
        out = np.zeros(tuple(int(x) for x in extent), dtype=dtype)
    assert out.dtype == np.dtype(dtype) and out.shape == tuple(int(x) for x in extent) and out.flags.c_contiguous
    fn = getattr(lib(), f"ndzip_oracle_decompress_u{bits}")
    n = fn(len(extent), _ext(extent), stream.ctypes.data, out.ctypes.data, int(num_threads))
    return out, int(n)


def forward_transform(cube_bits: np.ndarray, dims: int) -> np.ndarray:
    x = np.ascontiguousarray(cube_bits).copy()
    assert x.size == 4096
    getattr(lib(), f"ndzip_oracle_forward_transform_u{_bits(x.dtype)}")(x.ctypes.data, dims)
    return x


def inverse_transform(cube_bits: np.ndarray, dims: int) -> np.ndarray:
    x = np.ascontiguousarray(cube_bits).copy()
    assert x.size == 4096
    getattr(lib(), f"ndzip_oracle_inverse_transform_u{_bits(x.dtype)}")(x.ctypes.data, dims)
    return x


def transpose_bits(words: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(words)
    bits = _bits(x.dtype)
    assert x.size == bits
    out = np.zeros_like(x)
    getattr(lib(), f"ndzip_oracle_transpose_bits_u{bits}")(x.ctypes.data, out.ctypes.data)
    return out


def encode_cube(residuals: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(residuals)
    bits = _bits(x.dtype)
    assert x.size == 4096
    out = np.zeros(4096 + 4096 // bits, dtype=x.dtype)
    n = getattr(lib(), f"ndzip_oracle_encode_cube_u{bits}")(x.ctypes.data, out.ctypes.data)
    return out[:n].copy()


def decode_cube(stream: np.ndarray):
    x = np.ascontiguousarray(stream)
    bits = _bits(x.dtype)
    # pad so a malformed stream cannot read out of bounds
    padded = np.zeros(4096 + 4096 // bits, dtype=x.dtype)
    padded[: x.size] = x
    out = np.zeros(4096, dtype=x.dtype)
    n = getattr(lib(), f"ndzip_oracle_decode_cube_u{bits}")(padded.ctypes.data, out.ctypes.data)
    return out, int(n)


def load_cube(data: np.ndarray, hc: int) -> np.ndarray:
    data = np.ascontiguousarray(data)
    bits = _bits(data.dtype)
    out = np.zeros(4096, dtype=np.uint32 if bits == 32 else np.uint64)
    getattr(lib(), f"ndzip_oracle_load_cube_u{bits}")(len(data.shape), _ext(data.shape), data.ctypes.data, hc, out.ctypes.data)
    return out


def parallel_empty_like(a: np.ndarray, num_threads: int, copy: bool = True) -> np.ndarray:
    """A new array whose pages are first-touched by the OpenMP team (copy of `a`, or zeros) -- timing runs only."""
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    L = lib()
    L.ndzip_oracle_parallel_copy.restype = None
    L.ndzip_oracle_parallel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.ndzip_oracle_parallel_copy(out.ctypes.data, a.ctypes.data if copy else None, a.nbytes, int(num_threads))
    return out


def max_threads() -> int:
    return int(lib().ndzip_oracle_max_threads())


# ---------------------------------------------------------------------------------------------------------
# the real reference (only where oracle/_ref/libndzip_ref.so exists)
# ---------------------------------------------------------------------------------------------------------

def ref_compress(data: np.ndarray) -> np.ndarray:
    data = np.ascontiguousarray(data)
    bits = _bits(data.dtype)
    suf = "f32" if bits == 32 else "f64"
    wdt = np.uint32 if bits == 32 else np.uint64
    extent = data.shape
    bound = int(ref().ndzip_ref_compressed_length_bound(bits, len(extent), _ext(extent)))
    out = np.zeros(max(1, bound), dtype=wdt)  # zeroed: the CPU reference never writes the f64 header pad
    n = getattr(ref(), f"ndzip_ref_compress_{suf}")(len(extent), _ext(extent), data.ctypes.data, out.ctypes.data)
    if n < 0:
        raise RuntimeError("reference compress threw")
    return out[: int(n)].copy()


def ref_decompress(stream: np.ndarray, dtype, extent):
    bits = _bits(dtype)
    suf = "f32" if bits == 32 else "f64"
    stream = np.ascontiguousarray(stream)
    out = np.zeros(tuple(int(x) for x in extent), dtype=dtype)
    n = getattr(ref(), f"ndzip_ref_decompress_{suf}")(len(extent), _ext(extent), stream.ctypes.data, stream.size, out.ctypes.data)
    if n < 0:
        raise RuntimeError("reference decompress threw")
    return out, int(n)


def ref_compressed_length_bound(dtype, extent) -> int:
    return int(ref().ndzip_ref_compressed_length_bound(_bits(dtype), len(extent), _ext(extent)))
