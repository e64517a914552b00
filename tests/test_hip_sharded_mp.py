"""GPU test: the complete multi-process path of ShardedCodec (real torch.distributed collectives, asynchronous header
all-gather, fused base/offset kernel) with TWO ranks sharing the one GPU of the test box.  RCCL refuses two ranks on one
device, so the process group is gloo (it stages device tensors through the host); the calls ShardedCodec makes are the same."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ndzip_amd.sharded import assemble_stream, plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle

# Opt-in (NDZIP_TEST_SHARED_GPU=1): two processes time-sharing one GPU through gloo's host-staged collectives is not a
# supported deployment (one process per GPU is), and longer loops of it without host synchronisation have faulted / hung on
# the test box (known limitations) -- it must not be able to stall an unattended run of the GPU suite.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("NDZIP_TEST_SHARED_GPU") != "1", reason="opt-in: NDZIP_TEST_SHARED_GPU=1")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_MAIN = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ndzip_amd.sharded import ShardedCodec
from ndzip_amd.synth import synth_numpy

rank, world, out_dir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
extent = tuple(int(x) for x in sys.argv[5].split(","))
dtype = np.dtype(sys.argv[6]).type
dist.init_process_group("gloo", rank=rank, world_size=world)
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
full = synth_numpy(extent, dtype, seed=31, noise_mask=0xFF)
codec = ShardedCodec(dtype, extent, rank, world, device)
sh = codec.shard
slab = torch.from_numpy(np.ascontiguousarray(full[sh.start0: sh.start0 + sh.extent[0]])).to(device)
out = torch.zeros_like(slab)
for rep in range(3):          # repeated: header all-gather in flight across steps, handles reused
    codec.compress(slab)
    codec.decompress(out)
codec.check()
torch.cuda.synchronize()
ok = bool(torch.equal(out.view(torch.uint8), slab.view(torch.uint8)))
n = int(codec.body_len.cpu().numpy().view(np.uint32)[0])
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ok=ok, header=codec.header_global.cpu().numpy().view(np.uint32), body=codec.body[:n].cpu().numpy(), n=n,
         base=int(codec.base32.cpu().numpy().view(np.uint32)[0]))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extent,dtype", [((96, 64, 48), np.float32), ((256, 200), np.float64), ((50, 37, 41), np.float32)])
def test_two_processes_one_gpu(hiplib, cuda_device, tmp_path, extent, dtype):
    world = 2
    script = tmp_path / "rank_main.py"
    script.write_text(RANK_MAIN)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(tmp_path), ",".join(str(x) for x in extent),
                               np.dtype(dtype).name], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert all(bool(r["ok"]) for r in res)
    assert np.array_equal(res[0]["header"], res[1]["header"])           # every rank holds the whole global header
    shards = plan_shards(extent, world)
    full = synth_numpy(extent, dtype, seed=31, noise_mask=0xFF)
    got = assemble_stream(dtype, extent, res[0]["header"], [r["body"] for r in res], [int(r["n"]) for r in res], shards)
    want = oracle.compress(full)
    assert len(got) == len(want) and np.array_equal(got, want)
