"""bench.py on the CPU: the workload table matches BASELINE.json's configs, and the cpu_baseline legs (the genuine reference
codec on independent blocks, the OpenMP port, the serial reference) produce well-formed, round-trip-checked objects."""
import json
import os

import numpy as np
import pytest

import bench
from ndzip_amd.sharded import plan_shards
from ndzip_amd.synth import synth_numpy
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_table_covers_every_baseline_config():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        configs = json.load(f)["configs"]
    assert len(configs) == 5
    # per-GPU slabs times 8 GPUs reproduce configs[3] and configs[4]; config 5 is decompress-only
    _, dt4, slab4, _, mode4 = bench.CONFIGS["4"]
    assert (slab4[0] * 8,) + slab4[1:] == (2048, 1024, 1024) and dt4 == "float32" and mode4 == "both"
    _, dt5, slab5, _, mode5 = bench.CONFIGS["5"]
    assert (slab5[0] * 8,) + slab5[1:] == (1024, 1024, 1024) and dt5 == "float64" and mode5 == "decompress"
    assert bench.CONFIGS["1"][2] == (1 << 24,) and bench.CONFIGS["2"][2] == (512, 512, 512) and bench.CONFIGS["3"][2] == (8192, 8192)
    # the 16 GiB strong-scaling grid stays inside the format's uint32 counts and splits evenly over 1, 2, 4, 8 ranks
    g = bench.CONFIGS["16gib"][3]
    assert int(np.prod(g)) * 8 == 16 << 30 and int(np.prod(g)) < 2 ** 32
    for world in (1, 2, 4, 8):
        shards = plan_shards(g, world)
        assert len({s.num_hypercubes for s in shards}) == 1 and sum(s.num_hypercubes for s in shards) == int(np.prod(g)) // 4096
        assert shards[0].extent == (g[0] // world,) + g[1:]
    for world in (1, 2, 4, 8):  # cfg 4 / cfg 5 slabs at every scale the driver runs
        assert plan_shards((256 * world, 1024, 1024), world)[0].num_hypercubes == 65536
        assert plan_shards((128 * world, 1024, 1024), world)[0].num_hypercubes == 32768


@pytest.mark.parametrize("shape,threads", [((64, 64, 64), 8), ((32, 48, 16), 5), ((4096 * 5,), 3), ((256, 192), 7), ((16, 16, 16), 4)])
def test_blocks_partition_whole_hypercubes(shape, threads):
    blocks = bench._blocks(shape, threads)
    assert 1 <= len(blocks) <= threads
    side = {1: 4096, 2: 64, 3: 16}[len(shape)]
    seen = np.zeros(shape, dtype=np.int32)
    for b in blocks:
        assert all(s.start % side == 0 and s.stop % side == 0 for s in b)
        seen[b] += 1
    assert (seen == 1).all()


def _check_leg(leg, kind):
    assert set(leg) >= {"value", "unit", "cores", "kind", "sample"} and leg["unit"] == "GB/s" and leg["kind"] == kind
    assert leg["cores"] >= 1 and leg.get("roundtrip_ok", True)
    if leg["value"] is None:
        assert "reason" in leg  # a disturbed host: the field says so instead of a number
    else:
        assert leg["value"] > 0


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref is built from /root/reference (authoring container / shipped .so)")
def test_reference_legs_are_well_formed():
    grid = synth_numpy((64, 64, 64), np.float32, seed=1, noise_mask=0xFF)
    leg = bench.cpu_reference_blocks(grid, 4, budget_s=0.5)
    _check_leg(leg, "reference")
    assert leg["cores"] == 4 and len(leg["median_over_best"]) == 2
    s = bench.cpu_reference_serial(grid[:16], "test sample")
    _check_leg(s, "reference")
    assert s["cores"] == 1 and 0 < s["ratio"] < 1.1


def test_port_leg_is_well_formed():
    grid = synth_numpy((128, 128), np.float64, seed=2, noise_mask=0xFF)
    leg = bench.cpu_port_openmp(grid, 2, budget_s=0.5)
    _check_leg(leg, "port")
