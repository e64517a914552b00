"""Python host side of the MI355X back-end: a thin mirror of the reference's interfaces over the C ABI.

Reference interface (celerity/ndzip)                         here
---------------------------------------------------------------------------------------------------------
ndzip::compressed_length_bound<T>(extent)   ndzip.hh:224     compressed_length_bound(dtype, extent)
ndzip::compressor_requirements              ndzip.hh:255     CompressorRequirements
ndzip::make_cuda_compressor<T>(req, stream) cuda.hh:36       make_hip_compressor(dtype, req, stream)
ndzip::cuda_compressor<T>::compress         cuda.hh:18       HipCompressor.compress(in, extent, out, out_len)
ndzip::make_cuda_decompressor<T>(dims, s)   cuda.hh:40       make_hip_decompressor(dtype, dims, stream)
ndzip::cuda_decompressor<T>::decompress     cuda.hh:31       HipDecompressor.decompress(stream, out, extent)
ndzip::make_offloader<T>(target, dims)      offload.hh:59    make_hip_offloader(dtype, dims)
ndzip::offloader<T>::compress / decompress  offload.hh:16-24 HipOffloader.compress / .decompress

Everything here goes through ``libndzip_hip.so`` (include/ndzip_hip.h).  There is no CPU path: if the library
is missing or no GPU is visible the calls raise.  PyTorch is only used by callers for device memory and
streams; device buffers are passed as ``tensor.data_ptr()`` integers or tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libndzip_hip.so")  # (A/B tooling assigns this before the first lib() call; no environment override)
STAGES_LIB_PATH = os.path.join(_HERE, "libndzip_hip_stages.so")  # test hooks (debug_stage): never loaded by the product path

F32, F64 = 0, 1

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_DIMS_MISMATCH = -2
ERR_CAPACITY = -3
ERR_RUNTIME = -4
ERR_NO_DEVICE = -5
ERR_DEVICE_FAULT = -6
ERR_LIMIT = -7

_U32P = C.POINTER(C.c_uint32)
_lib = None
_stages_lib = None

# every symbol include/ndzip_hip.h declares (the CPU test suite checks the library exports all of them)
ABI_VERSION = 2  # NDZIP_HIP_ABI_VERSION of the include/ndzip_hip.h this binding was written against
EXPORTED_SYMBOLS = (
    "ndzip_hip_last_error",
    "ndzip_hip_abi_version",
    "ndzip_hip_device_info",
    "ndzip_hip_compressed_length_bound",
    "ndzip_hip_num_hypercubes",
    "ndzip_hip_header_words",
    "ndzip_hip_compressor_create",
    "ndzip_hip_compressor_compress",
    "ndzip_hip_compressor_compress_split",
    "ndzip_hip_compressor_offset_header",
    "ndzip_hip_compressor_offset_header_device",
    "ndzip_hip_compressor_offset_header_gathered",
    "ndzip_hip_compressor_check",
    "ndzip_hip_compressor_set_max_workgroups_per_cu",
    "ndzip_hip_compressor_destroy",
    "ndzip_hip_decompressor_create",
    "ndzip_hip_decompressor_decompress",
    "ndzip_hip_decompressor_decompress_bounded",
    "ndzip_hip_decompressor_decompress_split",
    "ndzip_hip_decompressor_decompress_split_bounded",
    "ndzip_hip_decompressor_set_f64_work_items",
    "ndzip_hip_decompressor_check",
    "ndzip_hip_decompressor_destroy",
    "ndzip_hip_offload_compress",
    "ndzip_hip_offload_decompress",
    "ndzip_hip_offloader_create",
    "ndzip_hip_offloader_destroy",
    "ndzip_hip_host_alloc",
    "ndzip_hip_host_free",
    "ndzip_hip_offloader_submit_compress",
    "ndzip_hip_offloader_submit_decompress",
    "ndzip_hip_offloader_wait",
    "ndzip_hip_stream_words",
    "ndzip_hip_chunked_plan",
    "ndzip_hip_chunked_compress",
    "ndzip_hip_chunked_decompress",
)
# include/ndzip_hip_stages.h: the parity tests' stage entry point lives in a library of its own (the product library holds no stage kernel)
STAGE_SYMBOLS = ("ndzip_hip_stages_last_error", "ndzip_hip_debug_stage", "ndzip_hip_debug_scratch_epoch_offset")


class NdzipHipError(RuntimeError):
    """std::runtime_error of the reference (cuda_check / dimensionality mismatch), with the C status code."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def lib():
    """Load libndzip_hip.so.  torch is imported first so the library binds to the HIP runtime torch already
    loaded (same SONAME libamdhip64.so.7) instead of bringing a second runtime into the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -m ndzip_amd.build` (there is no CPU fallback)")
    try:
        import torch  # noqa: F401  (side effect: loads torch's libamdhip64)
    except Exception:  # pragma: no cover - torch-less deployment uses the system runtime
        pass
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _bind_stages(L):
    L.ndzip_hip_stages_last_error.restype = C.c_char_p
    L.ndzip_hip_stages_last_error.argtypes = []
    L.ndzip_hip_debug_stage.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ndzip_hip_debug_stage.restype = C.c_int
    L.ndzip_hip_debug_scratch_epoch_offset.argtypes = []
    L.ndzip_hip_debug_scratch_epoch_offset.restype = C.c_uint32
    return L


def stages_lib():
    """libndzip_hip_stages.so (include/ndzip_hip_stages.h): loaded by the parity tests' debug_stage() only."""
    global _stages_lib
    if _stages_lib is not None:
        return _stages_lib
    if not os.path.exists(STAGES_LIB_PATH):
        raise FileNotFoundError(f"{STAGES_LIB_PATH} is missing: build it with `python -m ndzip_amd.build`")
    lib()  # (loads the HIP runtime the way the product library does)
    _stages_lib = _bind_stages(C.CDLL(STAGES_LIB_PATH))
    return _stages_lib


class _Tolerant:
    """A/B tooling only (bench.py --lib with a build of an EARLIER commit): entry points that build does not have yet bind to
    nothing instead of failing the load.  The package itself always binds strictly."""

    def __init__(self, L):
        object.__setattr__(self, "_L", L)

    def __getattr__(self, name):
        try:
            return getattr(self._L, name)
        except AttributeError:
            class _Missing:
                argtypes = None
                restype = None

                def __call__(self, *a):
                    raise NdzipHipError(ERR_RUNTIME, f"{name} is not exported by this build of the library")

            m = _Missing()
            object.__setattr__(self, name, m)
            return m


def _bind(L, strict: bool = True):
    """Declare the argument types of every entry point of include/ndzip_hip.h on a loaded library."""
    if not strict:
        L = _Tolerant(L)
    L.ndzip_hip_last_error.restype = C.c_char_p
    if strict:
        # (a library built from another revision of the header: an argument list may have changed under the same symbol name)
        try:
            L.ndzip_hip_abi_version.argtypes, L.ndzip_hip_abi_version.restype = [], C.c_int
            have = L.ndzip_hip_abi_version()
        except AttributeError:
            have = 1
        if have != ABI_VERSION:
            raise ImportError(f"libndzip_hip: ABI version {have}, this binding is written against {ABI_VERSION} (rebuild: python -m ndzip_amd.build)")
    L.ndzip_hip_device_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.ndzip_hip_compressed_length_bound.argtypes = [C.c_int, C.c_int, _U32P, C.POINTER(C.c_uint64)]
    L.ndzip_hip_num_hypercubes.argtypes = [C.c_int, _U32P, _U32P]
    L.ndzip_hip_header_words.argtypes = [C.c_int, C.c_uint32, _U32P]
    L.ndzip_hip_compressor_create.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
    L.ndzip_hip_compressor_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _U32P, C.c_void_p, C.c_void_p]
    L.ndzip_hip_compressor_compress_split.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _U32P, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ndzip_hip_compressor_offset_header.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.ndzip_hip_compressor_offset_header_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ndzip_hip_compressor_offset_header_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ndzip_hip_compressor_check.argtypes = [C.c_void_p]
    L.ndzip_hip_compressor_set_max_workgroups_per_cu.argtypes = [C.c_void_p, C.c_int]
    L.ndzip_hip_decompressor_set_f64_work_items.argtypes = [C.c_void_p, C.c_int]
    L.ndzip_hip_compressor_destroy.argtypes = [C.c_void_p]
    L.ndzip_hip_decompressor_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.ndzip_hip_decompressor_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _U32P]
    L.ndzip_hip_decompressor_decompress_split.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _U32P]
    L.ndzip_hip_decompressor_decompress_bounded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, _U32P]
    L.ndzip_hip_decompressor_decompress_split_bounded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, _U32P]
    L.ndzip_hip_decompressor_check.argtypes = [C.c_void_p]
    L.ndzip_hip_decompressor_destroy.argtypes = [C.c_void_p]
    L.ndzip_hip_offload_compress.argtypes = [C.c_int, C.c_int, _U32P, C.c_void_p, C.c_void_p, _U32P, C.POINTER(C.c_uint64)]
    L.ndzip_hip_offload_decompress.argtypes = [C.c_int, C.c_int, _U32P, C.c_void_p, C.c_uint32, C.c_void_p, _U32P, C.POINTER(C.c_uint64)]
    L.ndzip_hip_offloader_create.argtypes = [C.c_int, C.c_int, _U32P, C.c_int, C.POINTER(C.c_void_p)]
    L.ndzip_hip_offloader_destroy.argtypes = [C.c_void_p]
    L.ndzip_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.ndzip_hip_host_free.argtypes = [C.c_void_p]
    L.ndzip_hip_offloader_submit_compress.argtypes = [C.c_void_p, C.c_int, _U32P, C.c_void_p, C.c_void_p]
    L.ndzip_hip_offloader_submit_decompress.argtypes = [C.c_void_p, C.c_int, _U32P, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ndzip_hip_offloader_wait.argtypes = [C.c_void_p, C.c_int, _U32P, C.POINTER(C.c_uint64)]
    L.ndzip_hip_stream_words.argtypes = [C.c_int, C.c_int, _U32P, C.c_void_p, C.c_uint64, _U32P]
    _U64P = C.POINTER(C.c_uint64)
    L.ndzip_hip_chunked_plan.argtypes = [C.c_int, C.c_int, _U64P, C.c_uint64, _U64P, _U64P, _U64P]
    L.ndzip_hip_chunked_compress.argtypes = [C.c_int, C.c_int, _U64P, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, _U64P, _U64P]
    L.ndzip_hip_chunked_decompress.argtypes = [C.c_int, C.c_int, _U64P, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, _U64P, _U64P]
    for name in EXPORTED_SYMBOLS:
        if name != "ndzip_hip_last_error":
            getattr(L, name).restype = C.c_int
    return L


def _check(status: int) -> None:
    if status != OK:
        raise NdzipHipError(status, lib().ndzip_hip_last_error().decode() or f"ndzip_hip status {status}")


def _dtype_code(dtype) -> int:
    dt = np.dtype(dtype) if not _is_torch_dtype(dtype) else np.dtype(str(dtype).replace("torch.", ""))
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError(f"ndzip supports float32 and float64, not {dtype}")


def _is_torch_dtype(dtype) -> bool:
    return type(dtype).__module__ == "torch"


def word_dtype(dtype):
    return np.uint32 if _dtype_code(dtype) == F32 else np.uint64


def _ext(extent: Sequence[int]):
    extent = [int(x) for x in extent]
    if not 1 <= len(extent) <= 3:
        raise NdzipHipError(ERR_INVALID_ARGUMENT, "Invalid dimensionality")
    return (C.c_uint32 * 3)(*(extent + [1] * (3 - len(extent))))


def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    return int(x)


def device_info():
    arch = C.create_string_buffer(64)
    cus = C.c_int(0)
    _check(lib().ndzip_hip_device_info(arch, 64, C.byref(cus)))
    return arch.value.decode(), cus.value


def compressed_length_bound(dtype, extent) -> int:
    out = C.c_uint64(0)
    _check(lib().ndzip_hip_compressed_length_bound(_dtype_code(dtype), len(extent), _ext(extent), C.byref(out)))
    return out.value


def num_hypercubes(extent) -> int:
    out = C.c_uint32(0)
    _check(lib().ndzip_hip_num_hypercubes(len(extent), _ext(extent), C.byref(out)))
    return out.value


def header_words(dtype, nhc: int) -> int:
    out = C.c_uint32(0)
    _check(lib().ndzip_hip_header_words(_dtype_code(dtype), nhc, C.byref(out)))
    return out.value


class CompressorRequirements:
    """ndzip::compressor_requirements (ndzip.hh:255-269, common.cc:8-28): max hypercube count over extents
    of one dimensionality."""

    def __init__(self, *extents: Iterable[int]):
        self.dims = -1
        self.max_num_hypercubes = 0
        for e in extents:
            self.include(e)

    def include(self, extent) -> None:
        extent = tuple(int(x) for x in extent)
        if self.dims == -1:
            self.dims = len(extent)
        elif len(extent) != self.dims:
            raise RuntimeError(f"Cannot add a {len(extent)}-dimensional extent to {self.dims}-dimensional compressor_requirements")
        self.max_num_hypercubes = max(self.max_num_hypercubes, num_hypercubes(extent))


class HipCompressor:
    """Device-pointer compressor (ndzip::cuda_compressor<T>, cuda.hh:10-23).  Asynchronous on `stream`."""

    def __init__(self, dtype, requirements: CompressorRequirements, stream: int = 0):
        if requirements.dims == -1:
            raise RuntimeError("Cannot construct a compressor with empty requirements")  # common.hh:320
        self.dtype = np.dtype(word_dtype(dtype)).newbyteorder("=")
        self.code = _dtype_code(dtype)
        self.dims = requirements.dims
        h = C.c_void_p()
        _check(lib().ndzip_hip_compressor_create(self.code, self.dims, requirements.max_num_hypercubes, C.c_void_p(stream or None), C.byref(h)))
        self._h = h

    def compress(self, in_device_data, extent, out_device_stream, out_device_stream_length=None) -> None:
        _check(lib().ndzip_hip_compressor_compress(self._h, _ptr(in_device_data), len(extent), _ext(extent),
                                                    _ptr(out_device_stream), _ptr(out_device_stream_length)))

    def compress_split(self, in_device_data, extent, out_device_header, out_device_body, out_device_body_length=None) -> None:
        _check(lib().ndzip_hip_compressor_compress_split(self._h, _ptr(in_device_data), len(extent), _ext(extent),
                                                          _ptr(out_device_header), _ptr(out_device_body), _ptr(out_device_body_length)))

    def offset_header(self, device_header, count: int, base: int) -> None:
        _check(lib().ndzip_hip_compressor_offset_header(self._h, _ptr(device_header), count, base))

    def offset_header_device(self, device_header, count: int, device_base) -> None:
        _check(lib().ndzip_hip_compressor_offset_header_device(self._h, _ptr(device_header), count, _ptr(device_base)))

    def offset_header_gathered(self, device_header, count: int, device_lengths, device_borders, rank: int, world: int, device_base_out=None) -> None:
        _check(lib().ndzip_hip_compressor_offset_header_gathered(self._h, _ptr(device_header), count, _ptr(device_lengths),
                                                                 _ptr(device_borders), rank, world, _ptr(device_base_out)))

    def check(self) -> None:
        _check(lib().ndzip_hip_compressor_check(self._h))

    def set_max_workgroups_per_cu(self, n: int) -> None:
        """Cap the persistent compress grid at `n` workgroups per compute unit (0 = default: all that are resident)."""
        _check(lib().ndzip_hip_compressor_set_max_workgroups_per_cu(self._h, int(n)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ndzip_hip_compressor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipDecompressor:
    """Device-pointer decompressor (ndzip::cuda_decompressor<T>, cuda.hh:25-34)."""

    def __init__(self, dtype, dims: int, stream: int = 0):
        self.code = _dtype_code(dtype)
        self.dims = dims
        h = C.c_void_p()
        _check(lib().ndzip_hip_decompressor_create(self.code, dims, C.c_void_p(stream or None), C.byref(h)))
        self._h = h

    def decompress(self, in_device_stream, out_device_data, extent, stream_length_words: Optional[int] = None) -> None:
        """`stream_length_words` (not part of the reference interface): words the device stream holds; header entries that
        point past it are rejected on the device (error word) instead of being followed."""
        if stream_length_words is None:
            _check(lib().ndzip_hip_decompressor_decompress(self._h, _ptr(in_device_stream), _ptr(out_device_data), len(extent), _ext(extent)))
        else:
            _check(lib().ndzip_hip_decompressor_decompress_bounded(self._h, _ptr(in_device_stream), int(stream_length_words),
                                                                   _ptr(out_device_data), len(extent), _ext(extent)))

    def decompress_split(self, device_header, device_header_base, device_body, out_device_data, extent, body_words: Optional[int] = None) -> None:
        if body_words is None:
            _check(lib().ndzip_hip_decompressor_decompress_split(self._h, _ptr(device_header), _ptr(device_header_base), _ptr(device_body),
                                                                  _ptr(out_device_data), len(extent), _ext(extent)))
        else:
            _check(lib().ndzip_hip_decompressor_decompress_split_bounded(self._h, _ptr(device_header), _ptr(device_header_base),
                                                                         _ptr(device_body), int(body_words), _ptr(out_device_data),
                                                                         len(extent), _ext(extent)))

    def set_f64_work_items(self, n: int) -> None:
        """A/B switch: work-items per 64-bit hypercube -- 0 (default = 256), 128 or 256; same bits either way."""
        _check(lib().ndzip_hip_decompressor_set_f64_work_items(self._h, int(n)))

    def check(self) -> None:
        _check(lib().ndzip_hip_decompressor_check(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ndzip_hip_decompressor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipOffloader:
    """Host-pointer interface (ndzip::offloader<T>, offload.hh:8-34; behaviour of cuda_offloader,
    cuda_codec.inl:654-761).  `last_kernel_ns` is the reference's kernel_duration out-parameter."""

    def __init__(self, dtype, dims: int):
        self.np_dtype = np.dtype(dtype)
        self.code = _dtype_code(dtype)
        if not 1 <= dims <= 3:
            raise NdzipHipError(ERR_INVALID_ARGUMENT, "Invalid dimensionality")
        self.dims = dims
        self.last_kernel_ns = 0

    def compress(self, data: np.ndarray, extent=None) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=self.np_dtype)
        extent = tuple(data.shape) if extent is None else tuple(int(x) for x in extent)
        if len(extent) != self.dims:
            raise NdzipHipError(ERR_DIMS_MISMATCH, "data dimensionality does not match compressor dimensionality")
        bound = compressed_length_bound(self.np_dtype, extent)
        out = np.zeros(max(1, bound), dtype=word_dtype(self.np_dtype))
        n = C.c_uint32(0)
        ns = C.c_uint64(0)
        _check(lib().ndzip_hip_offload_compress(self.code, self.dims, _ext(extent), data.ctypes.data, out.ctypes.data, C.byref(n), C.byref(ns)))
        self.last_kernel_ns = ns.value
        return out[: n.value].copy()

    def decompress(self, stream: np.ndarray, extent):
        extent = tuple(int(x) for x in extent)
        if len(extent) != self.dims:
            raise NdzipHipError(ERR_DIMS_MISMATCH, "data dimensionality does not match decompressor dimensionality")
        stream = np.ascontiguousarray(stream, dtype=word_dtype(self.np_dtype))
        out = np.zeros(extent, dtype=self.np_dtype)
        n = C.c_uint32(0)
        ns = C.c_uint64(0)
        _check(lib().ndzip_hip_offload_decompress(self.code, self.dims, _ext(extent), stream.ctypes.data, stream.size,
                                                   out.ctypes.data, C.byref(n), C.byref(ns)))
        self.last_kernel_ns = ns.value
        return out, n.value


def stream_words(dtype, extent, stream: np.ndarray) -> int:
    """Length in words of the stream of an `extent` array that starts at stream[0], from its header alone."""
    extent = tuple(int(x) for x in extent)
    stream = np.ascontiguousarray(stream, dtype=word_dtype(dtype))
    n = C.c_uint32(0)
    _check(lib().ndzip_hip_stream_words(_dtype_code(dtype), len(extent), _ext(extent), stream.ctypes.data, stream.size, C.byref(n)))
    return n.value


def _ext64(extent):
    extent = [int(x) for x in extent]
    if not 1 <= len(extent) <= 3:
        raise NdzipHipError(ERR_INVALID_ARGUMENT, "Invalid dimensionality")
    return (C.c_uint64 * 3)(*(extent + [1] * (3 - len(extent))))


def chunked_plan(dtype, extent, max_elements: int = 0):
    """-> (rows_per_chunk, num_chunks, length_bound_words) of ndzip_hip_chunked_*: arrays beyond the format's uint32 counts are
    cut along dimension 0 into independent streams (the reference tool's multi-array file format, compress.cc:34-45)."""
    rows, n, bound = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _check(lib().ndzip_hip_chunked_plan(_dtype_code(dtype), len(extent), _ext64(extent), int(max_elements), C.byref(rows), C.byref(n), C.byref(bound)))
    return rows.value, n.value, bound.value


def chunked_compress(data: np.ndarray, max_elements: int = 0) -> np.ndarray:
    data = np.ascontiguousarray(data)
    _, _, bound = chunked_plan(data.dtype, data.shape, max_elements)
    out = np.zeros(max(1, bound), dtype=word_dtype(data.dtype))
    total, ns = C.c_uint64(0), C.c_uint64(0)
    _check(lib().ndzip_hip_chunked_compress(_dtype_code(data.dtype), data.ndim, _ext64(data.shape), int(max_elements), data.ctypes.data,
                                             out.ctypes.data, out.size, C.byref(total), C.byref(ns)))
    return out[: total.value].copy()


def chunked_decompress(streams: np.ndarray, dtype, extent, max_elements: int = 0):
    streams = np.ascontiguousarray(streams, dtype=word_dtype(dtype))
    out = np.zeros(tuple(int(x) for x in extent), dtype=dtype)
    consumed, ns = C.c_uint64(0), C.c_uint64(0)
    buf = streams if streams.size else np.zeros(1, dtype=streams.dtype)
    _check(lib().ndzip_hip_chunked_decompress(_dtype_code(dtype), len(extent), _ext64(extent), int(max_elements), buf.ctypes.data, streams.size,
                                               out.ctypes.data, C.byref(consumed), C.byref(ns)))
    return out, consumed.value


class PinnedBuffer:
    """Pinned host memory (ndzip_hip_host_alloc) exposed as a numpy array of `dtype`."""

    def __init__(self, nbytes: int, dtype=np.uint8):
        p = C.c_void_p()
        _check(lib().ndzip_hip_host_alloc(max(1, int(nbytes)), C.byref(p)))
        self._p = p
        self.nbytes = int(nbytes)
        raw = (C.c_uint8 * max(1, self.nbytes)).from_address(p.value)
        self.array = np.frombuffer(raw, dtype=np.uint8)[: self.nbytes].view(dtype)

    def close(self) -> None:
        if self._p is not None:
            self.array = None
            lib().ndzip_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipPipelinedOffloader:
    """Persistent host-pointer offloader with `slots` jobs in flight (ndzip_hip_offloader_*): the offloader<T>
    OBJECT of offload.hh:8-34 for callers that stream many arrays, e.g. the chunk loop of src/compress/compress.cc:17-86."""

    def __init__(self, dtype, max_extent, slots: int = 2):
        self.np_dtype = np.dtype(dtype)
        self.code = _dtype_code(dtype)
        self.max_extent = tuple(int(x) for x in max_extent)
        self.dims = len(self.max_extent)
        if not 1 <= self.dims <= 3:
            raise NdzipHipError(ERR_INVALID_ARGUMENT, "Invalid dimensionality")
        self.slots = int(slots)
        h = C.c_void_p()
        _check(lib().ndzip_hip_offloader_create(self.code, self.dims, _ext(self.max_extent), self.slots, C.byref(h)))
        self._h = h

    def _extent(self, extent):
        extent = tuple(int(x) for x in extent)
        if len(extent) != self.dims:
            raise NdzipHipError(ERR_DIMS_MISMATCH, "data dimensionality does not match compressor dimensionality")
        return extent

    def submit_compress(self, slot: int, data: np.ndarray, out_stream: np.ndarray, extent=None) -> None:
        extent = self._extent(data.shape if extent is None else extent)
        assert data.flags.c_contiguous and out_stream.flags.c_contiguous
        _check(lib().ndzip_hip_offloader_submit_compress(self._h, slot, _ext(extent), data.ctypes.data, out_stream.ctypes.data))

    def submit_decompress(self, slot: int, stream: np.ndarray, out_data: np.ndarray, extent=None) -> None:
        extent = self._extent(out_data.shape if extent is None else extent)
        assert stream.flags.c_contiguous and out_data.flags.c_contiguous
        _check(lib().ndzip_hip_offloader_submit_decompress(self._h, slot, _ext(extent), stream.ctypes.data, stream.size,
                                                           out_data.ctypes.data))

    def wait(self, slot: int):
        """-> (words, kernel_ns)"""
        n = C.c_uint32(0)
        ns = C.c_uint64(0)
        _check(lib().ndzip_hip_offloader_wait(self._h, slot, C.byref(n), C.byref(ns)))
        return n.value, ns.value

    def close(self) -> None:
        if self._h is not None:
            lib().ndzip_hip_offloader_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_hip_compressor(dtype, requirements, stream: int = 0) -> HipCompressor:
    if not isinstance(requirements, CompressorRequirements):
        requirements = CompressorRequirements(requirements)
    return HipCompressor(dtype, requirements, stream)


def make_hip_decompressor(dtype, dims: int, stream: int = 0) -> HipDecompressor:
    return HipDecompressor(dtype, dims, stream)


def make_hip_offloader(dtype, dims: int) -> HipOffloader:
    return HipOffloader(dtype, dims)


def debug_stage(stage: int, dtype, dims: int, extent, hc: int, d_in, d_out, d_out_len=None, n: int = 0, stream: int = 0) -> None:
    """Parity-test hook (include/ndzip_hip_stages.h, libndzip_hip_stages.so): one hypercube through one stage of the kernels."""
    ext = _ext(extent) if extent is not None else None
    L = stages_lib()
    status = L.ndzip_hip_debug_stage(stage, _dtype_code(dtype), dims, ext, hc, _ptr(d_in), _ptr(d_out), _ptr(d_out_len), n, C.c_void_p(stream or None))
    if status != OK:
        raise NdzipHipError(status, L.ndzip_hip_stages_last_error().decode() or f"ndzip_hip status {status}")
