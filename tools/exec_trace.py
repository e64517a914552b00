#!/usr/bin/env python
"""EXECUTED order of barriers, vector-memory operations and s_waitcnt vmcnt of ONE wavefront of a compress kernel, one iteration of
its persistent loop (CPU; test tooling: the built gfx950 code object on tests/gfx950_exec.py with a trace hook).  A listing's block
order is not execution order, and where hipcc places a wait -- or sinks a stretch of register-only work -- decides what a wavefront
overlaps with what; round 5 found a full memory drain in front of ~270 independent instructions this way (docs/rounds.md section 5).
Each line: +instructions executed since the previous line, address, instruction.
usage: exec_trace.py [--wave 0] [--f64] [--dims 3] [--decompress] [--lib ndzip_amd/_variants/<name>.so] [--lgkm]"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wave", type=int, default=0)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--lgkm", action="store_true", help="also list s_waitcnt lgkmcnt and LDS barriers' neighbours")
    ap.add_argument("--decompress", action="store_true", help="a decompress kernel instead (one workgroup start to end)")
    ap.add_argument("--f64-work-items", type=int, default=0)
    ap.add_argument("--dims", type=int, default=3, choices=[1, 2, 3])
    ap.add_argument("--ends", action="store_true", help="the prologue (kernel start to the first iteration's B1) and the drain (behind the last B4) instead")
    a = ap.parse_args()
    from ndzip_amd import hip
    from ndzip_amd.synth import synth_numpy
    from oracle import oracle
    from tests import gfx950_exec as gx
    from tests.wavesim import build as simbuild
    from tests.wavesim import sim

    named = gx.Bridge.kernel_named

    def with_lab_suffix(self, host_name):  # (lab builds: the kernels take one more uint32, ignored unless built for ablation)
        k = named(self, host_name)
        return k if k is not None else named(self, host_name + "j")

    gx.Bridge.kernel_named = with_lab_suffix
    log, count = [], [0]

    def trace(w, ins):
        if w.wg.index != 0 or w.index != a.wave:
            return
        count[0] += 1
        op = ins.op
        if op.startswith(("global_", "s_barrier", "s_sleep")) or (op == "s_waitcnt" and ("vmcnt" in ins.text or a.lgkm)):
            log.append((count[0], ins.addr, ins.text.strip()[:100]))

    b = gx.Bridge(simbuild.build(), [os.path.abspath(a.lib) if a.lib else hip.LIB_PATH], tempfile.mkdtemp(prefix="gfxtrace"))
    b.trace = trace
    b.only = ["decompress_kernel"] if a.decompress else ["compress_kernel_wide" if a.f64 else "compress_kernel_db"]
    # 64 tiles: 128 f32 hypercubes (two per tile) / 64 f64 hypercubes
    shape = {3: (64, 64, 128), 2: (512, 1024), 1: (128 * 4096,)}[a.dims] if not a.f64 else {3: (64, 64, 64), 2: (512, 512), 1: (64 * 4096,)}[a.dims]
    data = synth_numpy(shape, np.float64 if a.f64 else np.float32, seed=1, noise_mask=0xFF)
    if a.decompress:
        with b:
            back = sim.decompress(oracle.compress(data), data.dtype, data.shape, f64_work_items=a.f64_work_items or (128 if a.f64 else 0))
        assert np.array_equal(back.view(np.uint8), data.view(np.uint8))
        prev = 0
        for n, addr, text in log:
            print(f"+{n - prev:5d}  {addr:#07x}  {text}")
            prev = n
        print(f"+{count[0] - prev:5d}  (end)")
        return
    with b:
        got = sim.compress(data, cus=2, blocks_per_cu=2)  # 4 workgroups, 16 tiles each
    assert np.array_equal(got, oracle.compress(data))
    bars = [i for i, l in enumerate(log) if "s_barrier" in l[2]]
    if a.ends:
        last_b4 = bars[-2] if len(bars) % 4 == 2 else bars[-1]  # (the drain has one barrier of its own)
        for title, lo, hi in (("prologue", 0, bars[1]), ("drain", bars[-2], len(log) - 1)):
            print(f"## {title}")
            prev = log[lo][0] if lo else 0
            for n, addr, text in log[lo:hi + 1]:
                print(f"+{n - prev:5d}  {addr:#07x}  {text}")
                prev = n
        print(f"+{count[0] - prev:5d}  (end of the kernel)")
        return
    start, end = bars[1 + 4 * 2], bars[1 + 4 * 3]  # the prologue's barrier and two iterations skipped; four barriers per iteration
    prev = log[start][0]
    for n, addr, text in log[start:end + 1]:
        print(f"+{n - prev:5d}  {addr:#07x}  {text}")
        prev = n


if __name__ == "__main__":
    main()
