"""bench.main() with the kernels on the wave64 functional model and host tensors -- the entry script the CPU suite hands to
bench.py's own launcher (NDZIP_BENCH_ENTRY) so that `python bench.py --gpus 2 ...`, typed with no RANK / WORLD_SIZE in the
environment, can be run end to end where there is no GPU: torch.distributed.run starts the ranks, the group is gloo
(NDZIP_BENCH_SHARE_GPU=1), rank 0 prints the line.  Test infrastructure: numbers from such a run mean nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tests.test_bench_cpu import _HostAccelerator  # noqa: E402
from tests.wavesim import sim  # noqa: E402



def native_on_model():
    """--native-exchange here: sharded.cc compiled against the functional model, gloo behind the one-function collectives table."""
    import ctypes

    from ndzip_amd import sharded_native
    from tests.test_sharded_native_cpu import _gloo_table
    from tests.wavesim import build as simbuild

    sharded_native._lib = sharded_native._bind(ctypes.CDLL(simbuild.build_sharded()), rccl=False)
    real = sharded_native.NativeShardedCodec

    def with_gloo(dtype, extent, rank, world, device, **kw):
        table, _ = _gloo_table(world)
        return real(dtype, extent, rank, world, device, collectives=table, **kw)

    sharded_native.NativeShardedCodec = with_gloo


if __name__ == "__main__":
    os.environ["NDZIP_BENCH_SHARE_GPU"] = "1"
    bench.Accelerator = _HostAccelerator
    if "--native-exchange" in sys.argv:
        native_on_model()
    with sim.active():
        bench.main(sys.argv[1:])
