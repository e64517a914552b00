"""GPU parity against the committed golden fixtures (streams produced by the REAL reference, tests/golden/) and, at
BASELINE.json's full sizes, against the oracle plus size-independent properties (round trip, header monotonicity,
length bookkeeping)."""
import hashlib
import json
import os

import numpy as np
import pytest

from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import device_compress, device_decompress, same_bits

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLDEN, "hashes.json")) as f:
    META = json.load(f)


@pytest.mark.parametrize("case", META["small_cases"], ids=lambda c: c["name"])
def test_hip_reproduces_reference_streams(hiplib, cuda_device, case):
    vec = np.load(os.path.join(GOLDEN, "vectors.npz"))
    data, want = vec[case["name"] + "__in"], vec[case["name"] + "__stream"]
    got = device_compress(data)
    assert len(got) == len(want) and np.array_equal(got, want)
    assert same_bits(device_decompress(want, data.dtype, data.shape), data)


@pytest.mark.parametrize("case", META["hashed_cases"], ids=lambda c: f"{c['dtype']}-{'x'.join(map(str, c['shape']))}-{c['noise_mask']:#x}")
def test_hip_reproduces_reference_stream_hashes(hiplib, cuda_device, case):
    data = synth_numpy(case["shape"], np.dtype(case["dtype"]).type, case["seed"], case["noise_mask"])
    stream = device_compress(data)
    assert len(stream) == case["words"]
    assert hashlib.sha256(stream.tobytes()).hexdigest() == case["stream_sha256"]
    assert same_bits(device_decompress(stream, data.dtype, data.shape), data)


# every BASELINE.json config at full size; configs 4 and 5 as the z-slab one of the 8 ranks owns (the sharded path itself is
# tests/test_hip_sharded.py and the gloo test): 65 536 resp. 32 768 hypercubes, 1 GiB each
FULL = [("cfg2 3D f32 512^3", (512, 512, 512), np.float32), ("cfg3 2D f64 8192^2", (8192, 8192), np.float64),
        ("cfg1 1D f32 16Mi", (1 << 24,), np.float32), ("cfg4 rank slab 3D f32 256x1024x1024", (256, 1024, 1024), np.float32),
        ("cfg5 rank slab 3D f64 128x1024x1024", (128, 1024, 1024), np.float64)]


@pytest.mark.parametrize("name,shape,dtype", FULL, ids=[f[0] for f in FULL])
def test_full_size_configs(hiplib, cuda_device, name, shape, dtype):
    """BASELINE.json configs at full size: stream identical to the (multi-threaded) oracle, header strictly increasing and
    consistent with the stream length, decompress(compress(x)) == x."""
    import torch

    import ndzip_amd
    from ndzip_amd.synth import synth_torch

    tdt = torch.float32 if dtype == np.float32 else torch.float64
    d_in = synth_torch(shape, tdt, seed=1, noise_mask=0xFF, device=cuda_device)
    bound = ndzip_amd.compressed_length_bound(dtype, shape)
    wdt = torch.int32 if dtype == np.float32 else torch.int64
    d_stream = torch.zeros(bound, dtype=wdt, device=cuda_device)
    d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    comp = ndzip_amd.make_hip_compressor(dtype, ndzip_amd.CompressorRequirements(shape), torch.cuda.current_stream().cuda_stream)
    comp.compress(d_in, shape, d_stream, d_len)
    comp.check()
    n = int(d_len.cpu().numpy().view(np.uint32)[0])
    nhc = ndzip_amd.num_hypercubes(shape)
    hw = ndzip_amd.header_words(dtype, nhc)
    stream = d_stream[:n].cpu().numpy().view(np.uint32 if dtype == np.float32 else np.uint64)
    header = np.frombuffer(stream[:hw].tobytes(), dtype=np.uint32)[:nhc].astype(np.int64)
    per_hc = np.diff(np.concatenate([[0], header]))
    head_words = 4096 // (32 if dtype == np.float32 else 64)
    assert (per_hc >= head_words).all() and (per_hc <= 4096 + head_words).all()
    assert hw + header[-1] == n
    d_out = torch.empty_like(d_in)
    dec = ndzip_amd.make_hip_decompressor(dtype, len(shape), torch.cuda.current_stream().cuda_stream)
    dec.decompress(d_stream, d_out, shape)
    dec.check()
    it = torch.int32 if dtype == np.float32 else torch.int64
    assert torch.equal(d_out.view(it), d_in.view(it))
    host = d_in.cpu().numpy()
    want = oracle.compress(host, num_threads=oracle.max_threads())
    assert len(want) == n and np.array_equal(stream, want)
