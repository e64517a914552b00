"""The device-pointer entry points are plain stream work with NO per-launch state on the host (the compress kernels read the
descriptor epoch from their scratch and advance it themselves): a compress + decompress pair recorded into a hipGraph must replay
correctly on NEW data in the same buffers, any number of times.  (With the epoch as a kernel argument -- rounds 1-4 -- every replay
would have run under the recorded epoch, found the previous replay's tile descriptors "published" and summed their stale lengths.)
Reference counterpart: cuda_compressor::compress / cuda_decompressor::decompress are asynchronous on the caller's stream
(include/ndzip/cuda.hh:10-41) and therefore capturable; the reference's multi-kernel pipeline has no cross-launch state at all."""
import numpy as np
import pytest

from ndzip_amd.synth import synth_numpy
from oracle import oracle
from tests.util import same_bits

pytestmark = pytest.mark.gpu


def _recorded(fn, stream, rehearsal):
    """fn() recorded into a hipGraph on `stream`; returns the callable that replays it.  In the rehearsal on the functional model
    (no graphs there) the replay is fn itself: the C ABI called again with the very same arguments -- which is all a replay is, now
    that a launch has no state on the host."""
    if rehearsal:
        return fn
    import torch

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g.replay


@pytest.mark.parametrize("dtype,shapes", [(np.float32, [(48, 64, 64), (32, 32, 64)]), (np.float64, [(128, 192)]), (np.float32, [(4096 * 24,)])])
def test_recorded_compress_and_decompress_replay_on_new_data(hiplib, cuda_device, rehearsal, dtype, shapes):
    import torch

    import ndzip_amd

    t_dtype = torch.float32 if dtype == np.float32 else torch.float64
    wdt = torch.int32 if dtype == np.float32 else torch.int64
    side = torch.cuda.Stream(device=cuda_device)
    with torch.cuda.stream(side):
        comp = ndzip_amd.make_hip_compressor(dtype, ndzip_amd.CompressorRequirements(*shapes), side.cuda_stream)
        dec = ndzip_amd.make_hip_decompressor(dtype, len(shapes[0]), side.cuda_stream)
    graphs = []
    for shape in shapes:  # one graph per extent, all on the same handles (and therefore the same scratch and epoch)
        d_in = torch.zeros(shape, dtype=t_dtype, device=cuda_device)
        d_stream = torch.zeros(ndzip_amd.compressed_length_bound(dtype, shape), dtype=wdt, device=cuda_device)
        d_len = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        d_back = torch.zeros(shape, dtype=t_dtype, device=cuda_device)
        with torch.cuda.stream(side):  # warm-up outside the capture: module load, the cached occupancy query
            comp.compress(d_in, shape, d_stream, d_len)
            dec.decompress(d_stream, d_back, shape)
        side.synchronize()
        def both(shape=shape, d_in=d_in, d_stream=d_stream, d_len=d_len, d_back=d_back):
            comp.compress(d_in, shape, d_stream, d_len)
            dec.decompress(d_stream, d_back, shape)

        graphs.append((shape, _recorded(both, side, rehearsal), d_in, d_stream, d_len, d_back))
    torch.cuda.synchronize()
    for rep in range(6):
        shape, replay, d_in, d_stream, d_len, d_back = graphs[rep % len(graphs)]
        data = synth_numpy(shape, dtype, seed=100 + rep, noise_mask=0xFFFF if rep % 2 else 0xF)  # other lengths every replay
        want = oracle.compress(data, num_threads=oracle.max_threads())
        d_in.copy_(torch.from_numpy(data).to(cuda_device))
        d_back.zero_()
        torch.cuda.synchronize()
        replay()
        torch.cuda.synchronize()
        comp.check()
        dec.check()
        n = int(d_len.cpu().numpy().view(np.uint32)[0])
        got = d_stream[:n].cpu().numpy().view(want.dtype)
        assert n == len(want) and np.array_equal(got, want), f"replay {rep}"
        assert same_bits(d_back.cpu().numpy(), data), f"replay {rep}"
    comp.close()
    dec.close()
