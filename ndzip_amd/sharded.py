"""Multi-GPU sharding of the block codec: one process per GPU, contiguous hypercube ranges per rank.

The reference has no distributed runtime; this follows SURVEY.md section 8e:

  * the slowest dimension is cut into `world` slabs whose thickness is a multiple of the hypercube side, so a
    slab is a contiguous sub-array with a contiguous hypercube-index range and no halo;
  * every rank compresses its slab with LOCAL offsets (ndzip_hip_compressor_compress_split);
  * the only exchange is an all-gather of one uint32 body length per rank (-> exclusive prefix = the rank's global
    word offset, computed and added to the rank's header entries by ONE kernel) and an all-gather of the header
    segments after that.  Bodies never move:
    rank r's body lives at global body offset base_r, and the global stream is the concatenation
    [header][body_0]...[body_{R-1}][border_0]...[border_{R-1}] -- a host/file-level operation outside the
    timed region (`assemble_stream`), byte-identical to the single-GPU stream;
  * decompression needs no collective: a rank decodes its slab from its header slice, its base and its body -- or, while
    the exchange of the compress() before it is still in flight (`overlap_exchange`), from its local offsets, with the
    exchange running behind the decode kernel.

The exchange is written against torch.distributed only, so the same code runs over RCCL (backend "nccl",
device tensors) on the GPU box and over gloo (CPU tensors) in the world_size-2 CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

SIDE = {1: 4096, 2: 64, 3: 16}


@dataclass(frozen=True)
class Shard:
    rank: int
    start0: int          # first index along dimension 0
    extent: Tuple[int, ...]  # local extent (dimension 0 cut, others whole)
    hc_begin: int        # global hypercube index range [hc_begin, hc_end)
    hc_end: int
    border: int          # local border element count

    @property
    def num_hypercubes(self) -> int:
        return self.hc_end - self.hc_begin


def plan_shards(extent: Sequence[int], world: int) -> List[Shard]:
    """Cut dimension 0 into `world` slabs of whole hypercube planes; the last rank also takes the rows of
    dimension 0 that no hypercube covers.  Hypercube order is row-major over the hypercube grid
    (src/ndzip/common.hh:414-433), so a slab owns a contiguous index range."""
    extent = tuple(int(x) for x in extent)
    dims = len(extent)
    side = SIDE[dims]
    g = [e // side for e in extent]
    per_plane = 1
    for d in range(1, dims):
        per_plane *= g[d]
    whole_border = any(x == 0 for x in g)
    shards = []
    for r in range(world):
        p0 = r * g[0] // world
        p1 = (r + 1) * g[0] // world
        start = p0 * side
        stop = p1 * side if r + 1 < world else extent[0]
        local = (stop - start,) + extent[1:]
        n_local = 1
        for x in local:
            n_local *= x
        nhc = 0 if whole_border else (p1 - p0) * per_plane
        covered = nhc * 4096
        shards.append(Shard(r, start, local, p0 * per_plane if not whole_border else 0,
                            (p0 * per_plane if not whole_border else 0) + nhc, n_local - covered))
    return shards


def gather_headers(local_header, shard_sizes: Sequence[int], world: int, group=None, out=None):
    """All-gather the (already globalised) header segments into the full header on every rank.
    local_header: int32 tensor with this rank's entries.  Unequal segments are padded to the maximum.
    out: optional preallocated int32 tensor of world * max(shard_sizes) entries, used (and returned) when all segments are equal."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local_header.clone()
    m = max(shard_sizes)
    if m == 0:
        return local_header.clone()
    if all(s == m for s in shard_sizes):
        if out is None or out.numel() != world * m or out.dtype != local_header.dtype or out.device != local_header.device:
            out = torch.empty(world * m, dtype=local_header.dtype, device=local_header.device)
        dist.all_gather_into_tensor(out, local_header.contiguous(), group=group)
        return out
    padded = torch.zeros(m, dtype=local_header.dtype, device=local_header.device)
    padded[: local_header.numel()] = local_header
    out = torch.empty(world * m, dtype=local_header.dtype, device=local_header.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m: r * m + shard_sizes[r]] for r in range(world)])


INDEX_MAX = 0xFFFFFFFF  # index_type = uint32 (include/ndzip/ndzip.hh:20): element counts and stream offsets of the format


def base_from_lengths(lengths, borders, rank: int) -> int:
    """Global word offset of shard `rank`'s body: sum over lower ranks of (words written incl. border - border words).  Host
    restatement of offset_header_gathered_kernel (tests; the GPU path computes it in that kernel): like the kernel it sums ALL
    shards without wrapping and refuses (OverflowError; the kernel: error-word bit) a plan whose hypercube runs exceed the
    format's 32-bit offsets -- on every rank alike."""
    base = total = 0
    for r in range(len(lengths)):
        if r == rank:
            base = total
        total += (int(lengths[r]) & 0xFFFFFFFF) - int(borders[r])
    if total > INDEX_MAX:
        raise OverflowError(f"the hypercube runs of the {len(lengths)} shards add up to {total} words: more than the stream format's 32-bit offsets address")
    return base


def check_global_extent(dtype, extent: Sequence[int]) -> None:
    """What the stream format can carry at all: ValueError otherwise, naming the limit.  Eight legal slabs (each < 2^32 elements)
    can form a global array that is not.  Three separate uint32 quantities (include/ndzip/ndzip.hh:20): the element count; the
    header's offsets, which address hypercube runs only (src/ndzip/common.hh:351-358) -- held against the runs' bound, not the
    whole stream's; and the stream length word -- held against compressed_length_bound.  Host arithmetic only (the same three
    rules as libndzip_hip_rccl's check_global_extent, ndzip_amd/csrc/sharded.cc)."""
    import numpy as np

    extent = tuple(int(x) for x in extent)
    n = 1
    for x in extent:
        n *= x
    if n > INDEX_MAX:
        raise ValueError(f"extent {extent} has {n} elements: more than index_type (uint32) counts")
    bits = np.dtype(dtype).itemsize * 8
    side = SIDE[len(extent)]
    nhc = 1
    for x in extent:
        nhc *= x // side
    runs = nhc * (4096 + 4096 // bits)  # an incompressible hypercube: 4096 words + 4096 / B head words (common.cc:31-55)
    if runs > INDEX_MAX:
        raise ValueError(f"extent {extent}: {nhc} hypercubes can take {runs} words, more than the format's 32-bit offsets address")
    header = nhc if np.dtype(dtype).itemsize == 4 else (nhc + 1) // 2
    bound = header + runs + (n - nhc * 4096)
    if bound > INDEX_MAX:
        raise ValueError(f"extent {extent}: compressed_length_bound = {bound} words does not fit the uint32 stream length")


class ShardedCodec:
    """Per-rank driver of the sharded compress / decompress path on one GPU.

    All buffers are allocated once; compress() and decompress() enqueue work on the current torch stream and on the
    process group's stream only (no host synchronisation inside).  compress() = three steps with one collective between
    each: compress_local -> all-gather of one uint32 length per rank -> globalise (ONE kernel: base from the gathered
    lengths, added to the local header entries) -> all-gather of the header segments.  The steps are public so that a test
    can play every rank of a plan on one GPU."""

    def __init__(self, dtype, global_extent: Sequence[int], rank: int, world: int, device, group=None, async_header_gather: bool = False,
                 overlap_exchange: bool = False):
        import numpy as np
        import torch

        import ndzip_amd

        self.np_dtype = np.dtype(dtype)
        self.extent = tuple(int(x) for x in global_extent)
        check_global_extent(self.np_dtype, self.extent)
        self.dims = len(self.extent)
        self.rank, self.world, self.group = rank, world, group
        # True leaves the header all-gather in flight behind decompress (off until it has run on a multi-GPU node; the
        # default issues it synchronously on the process group's stream)
        self.async_header_gather = async_header_gather
        # True: compress() only STARTS the exchange (the all-gather of the body lengths, asynchronously); a decompress() that
        # follows decodes the slab from its LOCAL offsets -- it needs nothing from the other ranks -- and the rest of the exchange
        # (base + global offsets, header all-gather) is enqueued behind it, so both collectives run under the decode kernel.
        # Whoever reads header_global / base32, calls check() or compresses again completes the exchange first.
        self.overlap_exchange = overlap_exchange
        self._lengths_pending = None  # the length all-gather of the last compress(), not yet followed by globalise()
        self.device = device
        self.shards = plan_shards(self.extent, world)
        self.shard = self.shards[rank]
        self.words_per_elem_t = torch.int32 if self.np_dtype.itemsize == 4 else torch.int64
        # (a host `device` only occurs in the CPU test suite, which drives this class against the kernels' functional model)
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        self.compressor = ndzip_amd.make_hip_compressor(dtype, ndzip_amd.CompressorRequirements(self.shard.extent), stream)
        self.decompressor = ndzip_amd.make_hip_decompressor(dtype, self.dims, stream)
        nhc = self.shard.num_hypercubes
        bound = ndzip_amd.compressed_length_bound(dtype, self.shard.extent) - ndzip_amd.header_words(dtype, nhc)
        self.header_local = torch.zeros(max(1, nhc + 1), dtype=torch.int32, device=device)
        self.body = torch.zeros(max(1, bound), dtype=self.words_per_elem_t, device=device)
        self.body_len = torch.zeros(1, dtype=torch.int32, device=device)     # uint32 bits: words written incl. the local border
        self.base32 = torch.zeros(1, dtype=torch.int32, device=device)       # uint32 bits: global word offset of this body
        self.lens_all = torch.zeros(world, dtype=torch.int32, device=device)  # all-gathered body_len
        self.borders = torch.tensor([s.border for s in self.shards], dtype=torch.int64, device=device).to(torch.int32)
        self.sizes = [s.num_hypercubes for s in self.shards]
        self._header_global: Optional["torch.Tensor"] = None
        self._pending = None  # the header all-gather still in flight on the process group's stream

    # ---- the three steps of compress ----------------------------------------------------------------------------
    def compress_local(self, local_in, kernel_events=None) -> None:
        """local_in: this rank's slab (device tensor) -> header_local (LOCAL offsets), body, body_len.
        kernel_events: optional (start, stop) torch.cuda.Event pair recorded tightly around the codec launch."""
        if kernel_events:
            kernel_events[0].record()
        self.compressor.compress_split(local_in, self.shard.extent, self.header_local, self.body, self.body_len)
        if kernel_events:
            kernel_events[1].record()

    def globalise(self) -> None:
        """lens_all (every rank's body_len) -> header_local holds GLOBAL offsets, base32 this rank's base."""
        self.compressor.offset_header_gathered(self.header_local, self.shard.num_hypercubes, self.lens_all, self.borders, self.rank,
                                               self.world, self.base32)

    def finish(self) -> None:
        """Wait (stream-wise) for the header all-gather of the last compress()."""
        self._complete_exchange()
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    @property
    def header_global(self):
        """All NHC header entries with global offsets (complete once the last compress()'s all-gather has landed)."""
        self.finish()
        return self._header_global

    @header_global.setter
    def header_global(self, value) -> None:
        self._header_global = value

    def compress(self, local_in, kernel_events=None) -> None:
        """Afterwards: self.header_global (all NHC entries, global offsets), self.body / self.body_len (resident body + local
        border), self.base32 (global word offset).  The header all-gather is left in flight (it needs nothing from, and nothing
        that follows on this rank needs anything from it: this rank's own entries are already global in header_local); it is
        waited for by finish(), by reading header_global, and before the next compress() overwrites its input."""
        import torch
        import torch.distributed as dist

        sh = self.shard
        self.finish()
        self.compress_local(local_in, kernel_events)
        if self.world == 1:
            # single shard: local offsets are global offsets, nothing to exchange
            self._header_global = self.header_local[: sh.num_hypercubes]
            return
        if self.overlap_exchange:
            self._lengths_pending = dist.all_gather_into_tensor(self.lens_all, self.body_len, group=self.group, async_op=True)
            return
        dist.all_gather_into_tensor(self.lens_all, self.body_len, group=self.group)   # world x 4 bytes
        self._gather_headers()

    def _gather_headers(self) -> None:
        """lens_all has landed (or its wait is enqueued): base + global offsets, then the header all-gather."""
        import torch
        import torch.distributed as dist

        sh = self.shard
        self.globalise()
        m = max(self.sizes)
        if m > 0 and all(n == m for n in self.sizes) and (self.async_header_gather or self.overlap_exchange):
            if self._header_global is None or self._header_global.numel() != self.world * m:
                self._header_global = torch.empty(self.world * m, dtype=torch.int32, device=self.device)
            self._pending = dist.all_gather_into_tensor(self._header_global, self.header_local[:m], group=self.group, async_op=True)
        else:
            # (the buffer of the previous step is reused: no allocation per step on the collective's path)
            self._header_global = gather_headers(self.header_local[: sh.num_hypercubes], self.sizes, self.world, self.group,
                                                 out=self._header_global)

    def _complete_exchange(self) -> None:
        """overlap_exchange: the part of compress() that was left for later (no-op otherwise / when already done)."""
        if self._lengths_pending is not None:
            self._lengths_pending.wait()
            self._lengths_pending = None
            self._gather_headers()

    def decompress(self, local_out, kernel_events=None) -> None:
        """Decode this rank's slab from (its header entries with global offsets, its base, its resident body).  The entries
        are header_local after globalise() == header_global[hc_begin:hc_end]; the base stays on the device (`base32` == the
        previous shard's last header entry).  No collective and no dependence on the header all-gather."""
        sh = self.shard
        # kernel_events: optional (start, stop) pair recorded tightly around the decode launch -- what follows it on the stream in
        # the overlapped mode (the wait for the length all-gather, the globalise kernel) is not the decode kernel's time
        pending = self._lengths_pending is not None
        if kernel_events:
            kernel_events[0].record()
        # (pending: the exchange of the last compress() is still at its first step: header_local holds LOCAL offsets, base 0)
        self.decompressor.decompress_split(self.header_local, None if pending else self.base32, self.body, local_out, sh.extent)
        if kernel_events:
            kernel_events[1].record()
        if pending:
            self._complete_exchange()  # ... and continues behind the decode kernel

    def check(self) -> None:
        self.finish()
        self.compressor.check()
        self.decompressor.check()


def assemble_stream(dtype, extent, header_global, bodies, body_lens, shards: Sequence[Shard]):
    """Host-level concatenation into the reference's single stream (numpy; outside any timed region)."""
    import numpy as np

    wdt = np.uint32 if np.dtype(dtype).itemsize == 4 else np.uint64
    nhc = sum(s.num_hypercubes for s in shards)
    per = 1 if wdt == np.uint32 else 2
    hw = (nhc + per - 1) // per
    hdr = np.zeros(hw * per, dtype=np.uint32)
    hdr[:nhc] = np.asarray(header_global, dtype=np.uint32)[:nhc]
    parts = [hdr.view(wdt)]
    borders = []
    for s, body, n in zip(shards, bodies, body_lens):
        body = np.asarray(body).view(wdt)
        n = int(n)
        parts.append(body[: n - s.border])
        borders.append(body[n - s.border: n])
    return np.concatenate(parts + borders)
