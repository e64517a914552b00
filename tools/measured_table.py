#!/usr/bin/env python
"""After a GPU call of tools/gpu_batch.sh (first / explain): the "measured" cells of DESIGN.md section 5c's predicted
table, from the files the batch left in gpurun_out/ (or their copies in profiles/).  Prints markdown rows; nothing is written.
usage: tools/measured_table.py <dir> <tag>        e.g. tools/measured_table.py gpurun_out r05a"""
import json
import os
import re
import statistics
import sys


def read(path):
    return open(path).read() if os.path.exists(path) else ""


def ab_rows(text):
    """'round 1 main: compress_ms 0.19 frac 0.59 | decompress_ms 0.16 frac 0.7' lines -> {variant: (median compress us, median decompress us | None, n)}"""
    acc = {}
    for m in re.finditer(r"round \d+ (\S+): compress_ms ([\d.]+) frac [\d.]+(?: \| decompress_ms ([\d.]+))?", text):
        acc.setdefault(m.group(1), []).append((float(m.group(2)), float(m.group(3)) if m.group(3) else None))
    out = {}
    for v, rows in acc.items():
        c = statistics.median(r[0] for r in rows) * 1e3
        d = [r[1] for r in rows if r[1] is not None]
        out[v] = (c, statistics.median(d) * 1e3 if d else None, len(rows))
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    p = lambda name: os.path.join(d, f"{tag}_{name}")  # noqa: E731
    line = read(p("bench_n1.json")).strip().splitlines()
    if line:
        b = json.loads(line[-1])
        r = b["roofline"]
        print(f"bench line: value {b['value']} {b['unit']}; compress {r['launch_ms'] * 1e3:.1f} us = {r['achieved']} GB/s = frac {r['frac']}"
              f" (of the measured copy ceiling {r.get('frac_of_measured_copy')}); traffic {r.get('traffic')}; decompress "
              f"{(r.get('decompress') or {}).get('launch_ms', 0) * 1e3:.1f} us frac {(r.get('decompress') or {}).get('frac')}; "
              f"cpu_baseline {b.get('cpu_baseline', {}).get('value')} GB/s on {b.get('cpu_baseline', {}).get('cores')} cores ({b.get('cpu_baseline', {}).get('kind')})")
    names = {"main": "HEAD", "r05a": "HEAD before the post-B3 reordering (`r05a.so`)", "trearly": "transposes pinned in front of the look-back (`trearly.so`)",
             "cobatch2": "copy-out reads batched x2 (`cobatch2.so`)", "winpub": "window read behind the publish (`winpub.so`)",
             "wg3": "3 wavefronts per SIMD (`wg3.so`)", "r04": "round-4 library", "r03": "round-3 library", "r02": "round-2 library", "r01": "round-1 library",
             "plainloads": "default-policy input loads (`plainloads.so`)", "plain": "no inline assembly, no scalar pins (`plain.so`)"}
    for title, f in (("cfg 2 (3D f32 512^3)", "ab_variants.txt"), ("cfg 1 (1D f32 16 Mi)", "ab_variants_cfg1.txt"), ("cfg 3 (2D f64 8192^2)", "ab_variants_f64_2d.txt"),
                     ("3D f64 512^3", "ab_variants_f64_3d.txt")):
        rows = ab_rows(read(p(f)))
        if not rows:
            continue
        base = rows.get("main", (None,))[0]
        print(f"\n| {title}: launch, median of the interleaved rounds | compress us | vs HEAD | decompress us |\n|---|---|---|---|")
        for v, (c, dd, n) in rows.items():
            rel = f"{(c / base - 1) * 100:+.1f} %" if base else ""
            print(f"| {names.get(v, v)} | {c:.1f} | {rel} | {'' if dd is None else f'{dd:.1f}'} |")
    w = read(p("workgroups_per_cu.txt"))
    if w:
        print("\n| workgroups per CU (0 = default 4) | compress ms | frac |\n|---|---|---|")
        for m in re.finditer(r"workgroups per CU (\d+): compress_ms ([\d.]+) frac ([\d.]+)", w):
            print(f"| {m.group(1)} | {m.group(2)} | {m.group(3)} |")
    for f in ("configs.txt", "kernel_times_f64.txt", "phase_timing.txt"):
        t = read(p(f))
        if t:
            print(f"\n--- {f}\n{t.rstrip()}")


if __name__ == "__main__":
    main()
