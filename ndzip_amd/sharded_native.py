"""ctypes binding of libndzip_hip_rccl.so (include/ndzip_hip_sharded.h): the C++ host of the multi-GPU path.

`NativeShardedCodec` has the surface of `ndzip_amd.sharded.ShardedCodec` (compress / decompress / check on device tensors), but
the plan, the buffers, the overflow rule and the two collectives live in C++ and go over RCCL directly: Python only hands over
pointers.  The ncclComm_t is bootstrapped through the library's own helpers -- rank 0 draws the unique id, torch.distributed
(whatever group is up) carries the 128 bytes to the other ranks, every rank calls ncclCommInitRank.

There is no CPU fallback: without libndzip_hip_rccl.so (python -m ndzip_amd.build) this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

from . import hip

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libndzip_hip_rccl.so")
ABI_VERSION = 1  # NDZIP_HIP_SHARDED_ABI_VERSION this binding was written against

EXPORTED_SYMBOLS = (
    "ndzip_hip_sharded_abi_version",
    "ndzip_hip_sharded_last_error",
    "ndzip_hip_sharded_plan",
    "ndzip_hip_sharded_create",
    "ndzip_hip_sharded_create_with_collectives",
    "ndzip_hip_rccl_unique_id",
    "ndzip_hip_rccl_comm_create",
    "ndzip_hip_rccl_comm_destroy",
    "ndzip_hip_local_group_create",
    "ndzip_hip_local_group_destroy",
    "ndzip_hip_local_group_barrier",
    "ndzip_hip_sharded_create_local",
    "ndzip_hip_sharded_device_count",
    "ndzip_hip_sharded_set_device",
    "ndzip_hip_sharded_compress_host",
    "ndzip_hip_sharded_compress_local_host",
    "ndzip_hip_sharded_decompress_host",
    "ndzip_hip_sharded_shard",
    "ndzip_hip_sharded_compress",
    "ndzip_hip_sharded_compress_local",
    "ndzip_hip_sharded_exchange",
    "ndzip_hip_sharded_decompress",
    "ndzip_hip_sharded_header_global",
    "ndzip_hip_sharded_body",
    "ndzip_hip_sharded_stream_layout",
    "ndzip_hip_sharded_write_stream",
    "ndzip_hip_sharded_load",
    "ndzip_hip_sharded_check",
    "ndzip_hip_sharded_destroy",
)
# (the model build of the CPU tests holds sharded.cc only: no RCCL there)
RCCL_SYMBOLS = ("ndzip_hip_sharded_create", "ndzip_hip_rccl_unique_id", "ndzip_hip_rccl_comm_create", "ndzip_hip_rccl_comm_destroy")


class Shard(C.Structure):
    """ndzip_hip_shard"""
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("start0", C.c_uint32), ("extent", C.c_uint32 * 3), ("hc_begin", C.c_uint32),
                ("hc_end", C.c_uint32), ("border_elements", C.c_uint32), ("body_capacity_words", C.c_uint64)]


class StreamLayout(C.Structure):
    """ndzip_hip_stream_layout"""
    _fields_ = [(n, C.c_uint64) for n in ("header_words", "runs_offset_words", "runs_words", "border_offset_words", "border_words", "stream_words")]


ALL_GATHER_U32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ERROR_STRING = C.CFUNCTYPE(C.c_char_p, C.c_void_p, C.c_int)


class Collectives(C.Structure):
    """ndzip_hip_collectives: the exchange behind a table (a host with another transport than RCCL fills it in)."""
    _fields_ = [("ctx", C.c_void_p), ("all_gather_u32", ALL_GATHER_U32), ("error_string", ERROR_STRING)]


_lib = None


def _bind(L, rccl: bool = True):
    u32p, vp = C.POINTER(C.c_uint32), C.c_void_p
    L.ndzip_hip_sharded_abi_version.argtypes, L.ndzip_hip_sharded_abi_version.restype = [], C.c_int
    L.ndzip_hip_sharded_last_error.argtypes, L.ndzip_hip_sharded_last_error.restype = [], C.c_char_p
    have = L.ndzip_hip_sharded_abi_version()
    if have != ABI_VERSION:
        raise ImportError(f"libndzip_hip_rccl: ABI version {have}, this binding is written against {ABI_VERSION} (rebuild: python -m ndzip_amd.build)")
    sig = {
        "ndzip_hip_sharded_plan": [C.c_int, C.c_int, u32p, C.c_uint32, C.c_uint32, C.POINTER(Shard)],
        "ndzip_hip_sharded_create_with_collectives": [C.c_int, C.c_int, u32p, C.c_uint32, C.c_uint32, C.POINTER(Collectives), vp, C.POINTER(vp)],
        "ndzip_hip_local_group_create": [C.c_uint32, C.POINTER(vp)],
        "ndzip_hip_local_group_destroy": [vp],
        "ndzip_hip_local_group_barrier": [vp],
        "ndzip_hip_sharded_create_local": [C.c_int, C.c_int, u32p, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(vp)],
        "ndzip_hip_sharded_device_count": [C.POINTER(C.c_int)],
        "ndzip_hip_sharded_set_device": [C.c_int],
        "ndzip_hip_sharded_compress_host": [vp, vp],
        "ndzip_hip_sharded_compress_local_host": [vp, vp],
        "ndzip_hip_sharded_decompress_host": [vp, vp],
        "ndzip_hip_sharded_shard": [vp, C.POINTER(Shard)],
        "ndzip_hip_sharded_compress": [vp, vp],
        "ndzip_hip_sharded_compress_local": [vp, vp],
        "ndzip_hip_sharded_exchange": [vp],
        "ndzip_hip_sharded_decompress": [vp, vp],
        "ndzip_hip_sharded_header_global": [vp, C.POINTER(vp), u32p],
        "ndzip_hip_sharded_body": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
        "ndzip_hip_sharded_stream_layout": [vp, C.POINTER(StreamLayout)],
        "ndzip_hip_sharded_write_stream": [vp, vp, C.c_uint64, C.c_int],
        "ndzip_hip_sharded_load": [vp, vp, C.c_uint64],
        "ndzip_hip_sharded_check": [vp],
        "ndzip_hip_sharded_destroy": [vp],
    }
    if rccl:
        sig.update({
            "ndzip_hip_sharded_create": [C.c_int, C.c_int, u32p, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(vp)],
            "ndzip_hip_rccl_unique_id": [vp],
            "ndzip_hip_rccl_comm_create": [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
            "ndzip_hip_rccl_comm_destroy": [vp],
        })
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = argtypes, C.c_int
    return L


def lib():
    """Load libndzip_hip_rccl.so (after libndzip_hip.so and torch, so that it binds to the HIP runtime and the RCCL the process
    already has: same SONAMEs, libamdhip64.so.7 / librccl.so.1)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `python -m ndzip_amd.build` (there is no CPU fallback)")
    hip.lib()
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _check(status: int) -> None:
    if status != 0:
        raise hip.NdzipHipError(status, (lib().ndzip_hip_sharded_last_error() or b"").decode() or f"ndzip_hip_sharded status {status}")


def _ext(extent: Sequence[int]):
    extent = [int(x) for x in extent]
    if not 1 <= len(extent) <= 3:
        raise hip.NdzipHipError(hip.ERR_INVALID_ARGUMENT, "Invalid dimensionality")
    return (C.c_uint32 * 3)(*(extent + [0] * (3 - len(extent))))


def plan(dtype, global_extent: Sequence[int], rank: int, world: int) -> Shard:
    """ndzip_hip_sharded_plan: shard `rank` of `world` (host arithmetic; works without a GPU)."""
    sh = Shard()
    _check(lib().ndzip_hip_sharded_plan(hip._dtype_code(dtype), len(global_extent), _ext(global_extent), rank, world, C.byref(sh)))
    return sh


def rccl_comm_from_group(rank: int, world: int, group=None) -> int:
    """An ncclComm_t of this library's own for the ranks of `group` (torch.distributed carries the unique id, nothing else).
    The current device must already be this rank's."""
    import torch
    import torch.distributed as dist

    L = lib()
    ident = (C.c_char * 128)()
    if rank == 0:
        _check(L.ndzip_hip_rccl_unique_id(ident))
    box = [bytes(ident)]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    comm = C.c_void_p()
    _check(L.ndzip_hip_rccl_comm_create(C.create_string_buffer(box[0], 128), rank, world, C.byref(comm)))
    torch.cuda.synchronize()
    return comm.value


class NativeShardedCodec:
    """Per-rank driver of the sharded path in C++ (libndzip_hip_rccl.so).  `collectives`: a filled-in Collectives table instead of
    RCCL (then `comm` is ignored); `comm`: an existing ncclComm_t (integer) instead of bootstrapping one over `group`."""

    def __init__(self, dtype, global_extent: Sequence[int], rank: int, world: int, device, group=None, comm: Optional[int] = None,
                 collectives: Optional[Collectives] = None, overlap_exchange: bool = False):
        import numpy as np
        import torch

        L = lib()
        self.np_dtype = np.dtype(dtype)
        self.extent = tuple(int(x) for x in global_extent)
        self.dims = len(self.extent)
        self.rank, self.world, self.device = rank, world, device
        self._own_comm = None
        self._collectives = collectives  # (kept alive: the C side calls through its function pointers)
        # True: compress() is the codec launch alone; a decompress() that follows decodes the slab from its LOCAL offsets (it needs
        # nothing from the other ranks) and the exchange is enqueued BEHIND the decode kernel; whoever asks for the layout, the
        # stream or check() completes it first (the order ndzip_hip_sharded_compress_local / _decompress / _exchange of the C ABI)
        self.overlap_exchange = overlap_exchange and world > 1
        self._exchange_due = False
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        h = C.c_void_p()
        if collectives is not None:
            _check(L.ndzip_hip_sharded_create_with_collectives(hip._dtype_code(dtype), self.dims, _ext(self.extent), rank, world,
                                                               C.byref(collectives), C.c_void_p(stream or None), C.byref(h)))
        else:
            if comm is None and world > 1:
                comm = self._own_comm = rccl_comm_from_group(rank, world, group)
            _check(L.ndzip_hip_sharded_create(hip._dtype_code(dtype), self.dims, _ext(self.extent), rank, world, C.c_void_p(comm or None),
                                              C.c_void_p(stream or None), C.byref(h)))
        self._h = h
        sh = Shard()
        _check(L.ndzip_hip_sharded_shard(self._h, C.byref(sh)))
        self.shard_info = sh
        from .sharded import plan_shards  # the same plan, as the Python-side record bench.py and the tests read

        self.shards = plan_shards(self.extent, world)
        self.shard = self.shards[rank]
        got = (sh.start0, tuple(sh.extent[: self.dims]), sh.hc_begin, sh.hc_end, sh.border_elements)
        want = (self.shard.start0, self.shard.extent, self.shard.hc_begin, self.shard.hc_end, self.shard.border)
        if got != want:
            raise AssertionError(f"libndzip_hip_rccl and ndzip_amd.sharded.plan_shards disagree on rank {rank}'s shard: {got} vs {want}")

    # ---- the data path -------------------------------------------------------------------------------------------
    def compress(self, local_in, kernel_events=None) -> None:
        """kernel_events: optional (start, stop) pair recorded tightly around the codec launch (the exchange follows it)."""
        L = lib()
        self.finish()  # (one exchange per compress_local)
        if kernel_events:
            kernel_events[0].record()
        _check(L.ndzip_hip_sharded_compress_local(self._h, hip._ptr(local_in)))
        if kernel_events:
            kernel_events[1].record()
        if self.overlap_exchange:
            self._exchange_due = True
        else:
            _check(L.ndzip_hip_sharded_exchange(self._h))

    def finish(self) -> None:
        """The exchange of the last compress(), if it was left for later."""
        if self._exchange_due:
            self._exchange_due = False
            _check(lib().ndzip_hip_sharded_exchange(self._h))

    def decompress(self, local_out, kernel_events=None) -> None:
        if kernel_events:
            kernel_events[0].record()
        _check(lib().ndzip_hip_sharded_decompress(self._h, hip._ptr(local_out)))
        if kernel_events:
            kernel_events[1].record()
        self.finish()  # (overlapped mode: ... and continues behind the decode kernel)

    def check(self) -> None:
        self.finish()
        _check(lib().ndzip_hip_sharded_check(self._h))

    # ---- results ---------------------------------------------------------------------------------------------------
    def pointers(self):
        """(d_header_global, num_entries, d_body, d_body_length_words, d_base_words) as integers."""
        self.finish()
        hp, n = C.c_void_p(), C.c_uint32()
        _check(lib().ndzip_hip_sharded_header_global(self._h, C.byref(hp), C.byref(n)))
        b, bl, ba = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().ndzip_hip_sharded_body(self._h, C.byref(b), C.byref(bl), C.byref(ba)))
        return hp.value, n.value, b.value, bl.value, ba.value

    def stream_layout(self) -> StreamLayout:
        lay = StreamLayout()
        self.finish()
        _check(lib().ndzip_hip_sharded_stream_layout(self._h, C.byref(lay)))
        return lay

    def body_words(self) -> int:
        """Words of this rank's body (hypercube runs + border) after the last compress; synchronises."""
        lay = self.stream_layout()
        return int(lay.runs_words + lay.border_words)

    def write_stream(self, host_stream, with_header: bool) -> None:
        """host_stream: a writable numpy array of stream words (layout.stream_words of them or more)."""
        _check(lib().ndzip_hip_sharded_write_stream(self._h, host_stream.ctypes.data, host_stream.size, 1 if with_header else 0))

    def load(self, host_stream) -> None:
        _check(lib().ndzip_hip_sharded_load(self._h, host_stream.ctypes.data, host_stream.size))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ndzip_hip_sharded_destroy(self._h)
            self._h = None
        if self._own_comm:
            lib().ndzip_hip_rccl_comm_destroy(C.c_void_p(self._own_comm))
            self._own_comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
