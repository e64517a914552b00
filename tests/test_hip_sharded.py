"""GPU tests of the split-buffer entry points and the per-rank ShardedCodec driver (single rank here; the collective
logic is covered by tests/test_sharded_cpu.py with gloo)."""
import numpy as np
import pytest

from ndzip_amd.sharded import ShardedCodec, assemble_stream, plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import same_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extent,dtype", [((64, 48, 32), np.float32), ((130, 200), np.float64), ((50, 37, 41), np.float32), ((3 * 4096 + 5,), np.float64)])
def test_emulated_shards_concatenate_to_the_single_stream(hiplib, cuda_device, extent, dtype):
    """Compress every shard of a 3-way plan on this one GPU through compress_split, globalise the headers the way the
    ranks do, and check the concatenation against the single-stream oracle; then decode each shard from its slice."""
    import torch

    import ndzip_amd

    world = 3
    full = synth_numpy(extent, dtype, seed=5, noise_mask=0xFF)
    shards = plan_shards(extent, world)
    wdt = torch.int32 if dtype == np.float32 else torch.int64
    headers, bodies, lens = [], [], []
    base = 0
    comps = []
    for sh in shards:
        local = np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])
        d_in = torch.from_numpy(local).to(cuda_device)
        comp = ndzip_amd.make_hip_compressor(dtype, ndzip_amd.CompressorRequirements(sh.extent))
        bound = ndzip_amd.compressed_length_bound(dtype, sh.extent)
        d_hdr = torch.zeros(max(1, sh.num_hypercubes + 1), dtype=torch.int32, device=cuda_device)
        d_body = torch.zeros(max(1, bound), dtype=wdt, device=cuda_device)
        d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        comp.compress_split(d_in, sh.extent, d_hdr, d_body, d_len)
        comp.offset_header(d_hdr, sh.num_hypercubes, base)
        comp.check()
        n = int(d_len.cpu()[0])
        headers.append(d_hdr[: sh.num_hypercubes].cpu().numpy().view(np.uint32))
        bodies.append(d_body[:n].cpu().numpy())
        lens.append(n)
        base += n - sh.border
        comps.append((comp, d_hdr, d_body))
    header_global = np.concatenate(headers) if headers else np.zeros(0, np.uint32)
    got = assemble_stream(dtype, extent, header_global, bodies, lens, shards)
    want = oracle.compress(full)
    assert len(got) == len(want) and np.array_equal(got, want)
    # decode shard by shard from (header slice, device-resident base, body)
    d_hg = torch.from_numpy(header_global.view(np.int32).copy()).to(cuda_device) if len(header_global) else torch.zeros(1, dtype=torch.int32, device=cuda_device)
    for sh, (comp, d_hdr, d_body) in zip(shards, comps):
        dec = ndzip_amd.make_hip_decompressor(dtype, len(extent))
        n_local = int(np.prod(sh.extent))
        d_out = torch.zeros(max(1, n_local), dtype=wdt, device=cuda_device)
        hdr_slice = d_hg[sh.hc_begin: sh.hc_end] if sh.num_hypercubes else d_hg
        base_ptr = d_hg[sh.hc_begin - 1:] if sh.hc_begin > 0 else None
        dec.decompress_split(hdr_slice, base_ptr, d_body, d_out, sh.extent)
        dec.check()
        back = d_out[:n_local].cpu().numpy().view(dtype).reshape(sh.extent)
        assert same_bits(back, full[sh.start0: sh.start0 + sh.extent[0]])


def test_sharded_codec_single_rank(hiplib, cuda_device):
    import torch

    extent = (96, 80, 64)
    codec = ShardedCodec(np.float32, extent, 0, 1, cuda_device)
    data = synth_numpy(extent, np.float32, seed=9, noise_mask=0xFF)
    d_in = torch.from_numpy(data).to(cuda_device)
    d_out = torch.empty_like(d_in)
    codec.compress(d_in)
    codec.decompress(d_out)
    codec.check()
    assert torch.equal(d_out.view(torch.int32), d_in.view(torch.int32))
    n = int(codec.body_len.cpu()[0])
    got = assemble_stream(np.float32, extent, codec.header_global.cpu().numpy().view(np.uint32), [codec.body[:n].cpu().numpy()], [n], codec.shards)
    assert np.array_equal(got, oracle.compress(data))


@pytest.mark.parametrize("extent,dtype,world", [((96, 64, 48), np.float32, 3), ((256, 200), np.float64, 4), ((8 * 4096 + 3,), np.float32, 2),
                                                  ((50, 37, 41), np.float32, 2)])
def test_every_rank_of_a_plan_on_one_gpu(hiplib, cuda_device, extent, dtype, world):
    """The N > 1 path of ShardedCodec, step by step, with the two collectives replaced by their definition (concatenate what
    every rank contributes): compress_local on every rank, all-gather of the lengths, globalise (the fused base + offset
    kernel, checked against its host restatement), all-gather of the header segments, assemble, decode every slab."""
    import torch

    from ndzip_amd.sharded import base_from_lengths

    full = synth_numpy(extent, dtype, seed=21, noise_mask=0xFF)
    codecs = [ShardedCodec(dtype, extent, r, world, cuda_device) for r in range(world)]
    slabs = [torch.from_numpy(np.ascontiguousarray(full[c.shard.start0: c.shard.start0 + c.shard.extent[0]])).to(cuda_device) for c in codecs]
    for rep in range(2):  # twice: the handles are reused (descriptor epochs, ticket counters)
        for c, slab in zip(codecs, slabs):
            c.compress_local(slab)
        lens_all = torch.cat([c.body_len for c in codecs])            # == all_gather_into_tensor(lens_all, body_len)
        for c in codecs:
            c.lens_all.copy_(lens_all)
            c.globalise()
        segments = [c.header_local[: c.shard.num_hypercubes] for c in codecs]
        header_global = torch.cat(segments)                            # == gather_headers(...)
        for c in codecs:
            c.header_global = header_global
            c.check()
            # a rank decodes from its own entries: they must be exactly its slice of the global header
            assert torch.equal(c.header_local[: c.shard.num_hypercubes], header_global[c.shard.hc_begin: c.shard.hc_end])
        lens = lens_all.cpu().numpy().view(np.uint32)
        borders = [c.shard.border for c in codecs]
        for r, c in enumerate(codecs):
            assert int(c.base32.cpu().numpy().view(np.uint32)[0]) == base_from_lengths(lens, borders, r)
        got = assemble_stream(dtype, extent, header_global.cpu().numpy().view(np.uint32), [c.body.cpu().numpy() for c in codecs],
                              [int(x) for x in lens], codecs[0].shards)
        want = oracle.compress(full)
        assert len(got) == len(want) and np.array_equal(got, want)
        for c, slab in zip(codecs, slabs):
            out = torch.zeros_like(slab)
            c.decompress(out)
            c.check()
            assert torch.equal(out.view(torch.uint8), slab.view(torch.uint8))
