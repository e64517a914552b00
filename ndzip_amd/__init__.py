"""ndzip_amd -- MI355X (gfx950) back-end for ndzip's block encode/decode path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (built in-tree into libndzip_hip.so by ndzip_amd.build)
  hip.py       Python mirror of the reference's compressor / decompressor / offloader interfaces (ctypes)
  cli/         `ndzip-hip`: file-level compress / decompress tool (the reference's src/compress for this back-end)
  sharded.py   multi-GPU hypercube-range sharding (one process per GPU, RCCL only for offsets + headers), driven through torch.distributed
  sharded_native.py   the same path through its C++ host (csrc/sharded.cc + sharded_rccl.cc -> libndzip_hip_rccl.so, include/ndzip_hip_sharded.h)
               `ndzip-hip-sharded` (cli/ndzip_hip_sharded_cli.cc): one array -> one stream over N ranks / GPUs
  synth.py     deterministic integer-only synthetic grids
"""
from .hip import (  # noqa: F401
    CompressorRequirements,
    HipCompressor,
    HipDecompressor,
    HipOffloader,
    HipPipelinedOffloader,
    PinnedBuffer,
    NdzipHipError,
    compressed_length_bound,
    device_info,
    header_words,
    make_hip_compressor,
    make_hip_decompressor,
    make_hip_offloader,
    num_hypercubes,
    stream_words,
    word_dtype,
)
