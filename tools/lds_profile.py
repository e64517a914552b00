#!/usr/bin/env python
"""LDS bank-conflict profile of the kernels WITHOUT a GPU (test tooling; uses tests/wavesim, never the product path).

The functional model's `ldsprof` variant hooks every load and store of the unchanged kernel sources, groups the LDS accesses of
a wavefront into wave-instructions and prices them with the MI355X guide's LDS model (tests/wavesim/wavesim.cc).  Bank conflicts
are a pure function of the addresses, so the two sums it reports are the offline counterparts of the hardware counters
  SQ_LDS_IDX_ACTIVE     = LDS-array cycles            ("cycles")
  SQ_LDS_BANK_CONFLICT  = cycles - conflict-free ones ("extra")
per static access (file:line of the kernel source, inlined frames included), which the PMC cannot give.

usage: lds_profile.py [--shape 128,128,128] [--dtype float32] [--data synth|random|zeros] [--top 25] [--mode compress|decompress|both]
Access widths are the host compiler's (-O1): a pair of adjacent 4-byte accesses that hipcc merges into one ds_read2_b32 / b64 is
priced as two b32 here -- same array cycles for read2_b32, an upper bound for b64."""
import argparse
import collections
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def symbolize(lib, offsets):
    """{offset: "file:line <- file:line ..."} innermost frame first, product sources only"""
    p = subprocess.run([SYMBOLIZER, "--obj=" + lib, "--output-style=JSON", "--inlines", "--relativenames"] + [hex(o - 1) for o in offsets],
                       capture_output=True, text=True, check=True)
    out = {}
    for off, entry in zip(offsets, json.loads(p.stdout)):
        frames = entry.get("Symbol", [])
        names = []
        for fr in frames:
            f = os.path.basename(fr.get("FileName", "?"))
            if f.endswith((".hpp", ".inl", ".hip")):
                names.append(f"{f}:{fr.get('Line', 0)}")
        out[off] = " <- ".join(names[:3]) if names else "?"
    return out


def run(shape, dtype, data, mode, f64_work_items=0):
    from ndzip_amd import hip, synth
    from tests.wavesim import sim

    dt = np.dtype(dtype).type
    if data == "random":
        rng = np.random.default_rng(1)
        x = rng.integers(0, 2 ** (8 * np.dtype(dt).itemsize) - 1, size=shape, dtype=np.uint32 if dt == np.float32 else np.uint64).view(dt)
    elif data == "zeros":
        x = np.zeros(shape, dtype=dt)
    else:
        x = synth.synth_numpy(shape, dt)
    L = sim.load("ldsprof")
    raw = L._lib if hasattr(L, "_lib") else L
    dump = raw.wavesim_lds_profile_dump if hasattr(raw, "wavesim_lds_profile_dump") else C.CDLL(raw._name).wavesim_lds_profile_dump
    dump.argtypes, dump.restype = [C.c_char_p], C.c_int
    results = {}
    with sim.active(4, 3, variant="ldsprof"):
        W = np.uint32 if dt == np.float32 else np.uint64
        bound = hip.compressed_length_bound(x.dtype, x.shape)
        out = np.zeros(bound + 8, dtype=W)
        length = np.zeros(1, dtype=np.uint32)
        comp = hip.make_hip_compressor(x.dtype, hip.CompressorRequirements(x.shape))
        comp.compress(x.ctypes.data, x.shape, out.ctypes.data, length.ctypes.data)
        comp.check()
        with tempfile.NamedTemporaryFile(suffix=".jsonl") as f:
            assert dump(f.name.encode()) == 0
            results["compress"] = [json.loads(l) for l in open(f.name)]
        y = np.empty_like(x)
        dec = hip.make_hip_decompressor(x.dtype, x.ndim)
        if f64_work_items:
            dec.set_f64_work_items(f64_work_items)
        dec.decompress(out.ctypes.data, y.ctypes.data, x.shape)
        dec.check()
        assert np.array_equal(y.view(W), x.view(W))
        with tempfile.NamedTemporaryFile(suffix=".jsonl") as f:
            assert dump(f.name.encode()) == 0
            results["decompress"] = [json.loads(l) for l in open(f.name)]
    ratio = int(length[0]) * x.itemsize / x.nbytes
    return {k: v for k, v in results.items() if mode in (k, "both")}, ratio


def report(name, rows, top, nhc):
    if not rows:
        print(f"## {name}: no LDS accesses recorded")
        return
    lib = rows[0]["lib"]
    sym = symbolize(lib, sorted({r["offset"] for r in rows}))
    agg = collections.OrderedDict()
    for r in rows:
        key = (sym[r["offset"]], r["bytes"], r["store"])
        a = agg.setdefault(key, dict(instructions=0, cycles=0, ideal=0, lanes=0, busy=0))
        for k in a:
            a[k] += r[k]
    tot_c = sum(a["cycles"] for a in agg.values())
    tot_i = sum(a["ideal"] for a in agg.values())
    issue = {(4, 0): 2, (8, 0): 2, (16, 0): 4, (4, 1): 4, (8, 1): 6, (16, 1): 13}
    for (where, size, store), a in agg.items():
        a["floor"] = a["instructions"] * issue.get((size, store), 2)  # what the instructions cost without any conflict
    tot_b = sum(a["busy"] for a in agg.values())
    tot_f = sum(a["floor"] for a in agg.values())
    print(f"## {name}: LDS-array cycles {tot_c} ({tot_c / nhc:.0f} per hypercube), conflict-free {tot_i}, extra {tot_c - tot_i} = "
          f"{(tot_c - tot_i) / tot_c:.1%} of the cycles  [hardware: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE]")
    print(f"   LDS-pipe busy cycles (max of array and issue cycles per instruction) {tot_b / nhc:.0f} per hypercube, {tot_f / nhc:.0f} without "
          f"conflicts: conflicts COST {(tot_b - tot_f) / nhc:.0f} cycles per hypercube ({(tot_b - tot_f) / tot_b:.1%})")
    print(f"{'cost/hc':>8} {'extra/hc':>9} {'cyc/hc':>8} {'x ideal':>7} {'instr/hc':>8} {'lanes':>5}  access")
    for (where, size, store), a in sorted(agg.items(), key=lambda kv: -(kv[1]["busy"] - kv[1]["floor"]))[:top]:
        kind = f"{'ds_write' if store else 'ds_read'}_b{8 * size}"
        print(f"{(a['busy'] - a['floor']) / nhc:8.1f} {(a['cycles'] - a['ideal']) / nhc:9.1f} {a['cycles'] / nhc:8.1f} {a['cycles'] / a['ideal']:7.2f} "
              f"{a['instructions'] / nhc:8.1f} {a['lanes'] / a['instructions']:5.1f}  {kind:14s} {where}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="128,128,128")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--data", default="synth", choices=["synth", "random", "zeros"])
    ap.add_argument("--mode", default="both", choices=["compress", "decompress", "both"])
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--f64-work-items", type=int, default=0, help="64-bit decoder: 0 (the library's default = 256), 128 or 256 work-items per hypercube")
    a = ap.parse_args()
    shape = tuple(int(s) for s in a.shape.split(","))
    results, ratio = run(shape, a.dtype, a.data, a.mode, a.f64_work_items)
    from ndzip_amd import hip

    nhc = hip.num_hypercubes(shape)
    print(f"# LDS profile on the functional model: {a.dtype} {'x'.join(map(str, shape))} ({a.data}), {nhc} hypercubes, compression ratio {ratio:.3f}")
    for name, rows in results.items():
        report(name, rows, a.top, nhc)


if __name__ == "__main__":
    main()
