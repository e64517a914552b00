#!/usr/bin/env python
"""Executed-instruction profile of the BUILT kernels WITHOUT a GPU (test tooling; tests/gfx950_exec.py interprets the gfx950 code objects):
wave-instructions actually executed per hypercube, by class, for the compress and the decompress kernel of a profile on the
benchmark's synthetic data -- the dynamic counterpart of tools/isa_cost.py, whose static counts include both sides of every branch
(dense-chunk paths, look-back retry loops, the drain).  VALU issue cycles per hypercube and SIMD = VALU x 4 / 4 SIMDs x wavefronts.

usage: dynamic_profile.py [--dtype float32] [--shape 32,64,64] [--noise-mask 0xff] [--f64-work-items 0|128|256]"""
import argparse
import collections
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def klass(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_"):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep"):
        return op
    return "SALU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--shape", default="32,64,64")
    ap.add_argument("--noise-mask", type=lambda s: int(s, 0), default=0xFF)
    ap.add_argument("--f64-work-items", type=int, default=0)
    ap.add_argument("--lib", default=None, help="a variant library (ndzip_amd/_variants/<name>.so) instead of the product library")
    ap.add_argument("--scalar", action="store_true", help="also list every scalar-side opcode executed (SALU / waits / branches)")
    a = ap.parse_args()
    from ndzip_amd import hip, synth
    from oracle import oracle
    from tests import gfx950_exec as gx
    from tests.wavesim import build as simbuild
    from tests.wavesim import sim

    shape = tuple(int(x) for x in a.shape.split(","))
    dt = np.dtype(a.dtype).type
    if dt == np.float64 and not a.f64_work_items:
        a.f64_work_items = 128  # (the library's default; named explicitly, or the test helper would run every mapping for comparison)
    data = synth.synth_numpy(shape, dt, seed=1, noise_mask=a.noise_mask)
    want = oracle.compress(data)
    nhc = hip.num_hypercubes(shape)
    gx.PROFILE = collections.Counter()
    gx.LDS_PROFILE = collections.defaultdict(lambda: [0, 0, 0, 0])
    if a.lib:  # (lab builds: the kernels take one more uint32, the experiment flags)
        orig = gx.Bridge.kernel_named
        gx.Bridge.kernel_named = lambda self, host_name: orig(self, host_name) or orig(self, host_name + "j")
    bridge = gx.Bridge(simbuild.build(), [os.path.abspath(a.lib) if a.lib else hip.LIB_PATH], tempfile.mkdtemp(prefix="gfx950_prof"))
    with bridge:
        got = sim.compress(data, cus=4, blocks_per_cu=2)
        back = sim.decompress(want, dt, shape, f64_work_items=a.f64_work_items)
    assert np.array_equal(got, want) and np.array_equal(back.view(np.uint8), data.view(np.uint8))
    ratio = want.nbytes / data.nbytes
    print(f"# executed wave-instructions per hypercube: {a.dtype} {'x'.join(map(str, shape))}, {nhc} hypercubes, ratio {ratio:.3f}"
          + (f", f64 decoder with {a.f64_work_items} work-items" if a.f64_work_items else ""))
    for name, grid, block, total in bridge.launched:
        if "border" in name:
            continue
        per = collections.Counter()
        for (k, op), n in gx.PROFILE.items():
            if k == name or k == name + "j":
                per[klass(op)] += n
        short = name.split("N_1")[1][2:44]
        waves_per_hc = {True: 4, False: 2}["wide" in name] if "decompress" in name else (4 if "wide" in name else 2)
        valu = per["VALU"] / nhc
        print(f"{short:44s} grid {grid:3d} x {block}: " + "  ".join(f"{c} {per[c] / nhc:7.1f}" for c in ("VALU", "SALU", "LDS", "VMEM", "branch", "s_waitcnt", "s_nop", "s_barrier"))
              + f"  | per wavefront: VALU {valu / waves_per_hc:6.0f}  | VALU issue cycles per hypercube and SIMD (4 SIMDs share a hypercube's wavefronts): {valu * 4 / 4:6.0f}")
        lds = [(op, r) for (k, op), r in gx.LDS_PROFILE.items() if k in (name, name + "j")]
        arr, ideal, busy = (sum(r[j] for _, r in lds) for j in (1, 2, 3))
        floor = sum(r[0] * gx._LDS_RULES[op][2] for op, r in lds)
        print(f"    LDS (the guide's lane groups and banks on the executed addresses): array cycles {arr / nhc:.0f} per hypercube, conflict-free {ideal / nhc:.0f}, "
              f"extra {(arr - ideal) / max(arr, 1):.1%} [SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE]; pipe busy {busy / nhc:.0f}, {floor / nhc:.0f} without conflicts; "
              + ", ".join(f"{op} {r[1] / max(r[2], 1):.2f}x" for op, r in sorted(lds, key=lambda kv: -(kv[1][1] - kv[1][2]))[:5]))
        top = collections.Counter({op: n for (k, op), n in gx.PROFILE.items() if k in (name, name + "j")})
        print("    top: " + ", ".join(f"{op} {n / nhc:.0f}" for op, n in top.most_common(14)))
        if a.scalar:
            print("    scalar side: " + ", ".join(f"{op} {n / nhc:.0f}" for op, n in top.most_common() if op.startswith("s_")))


if __name__ == "__main__":
    main()
