#!/usr/bin/env python
"""Compact trace of memory ops / barriers of one kernel with the highest VGPR index each touches -- shows where
register pressure peaks.  usage: isa_trace.py file.s <kernel name substring>"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M):
    if key not in m.group(1):
        continue
    body = m.group(2).split("\n")

    def maxv(line):
        mx = -1
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
            mx = max(mx, int(b))
        for a in re.findall(r"\bv(\d+)\b", line):
            mx = max(mx, int(a))
        return mx

    # running max over all instructions between "interesting" ops
    prev = None
    run_max = -1
    n_valu = 0
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t[0] in ".;":
            continue
        op = t.split()[0]
        run_max = max(run_max, maxv(t))
        if op.startswith("v_"):
            n_valu += 1
        if op.startswith(("ds_", "global_", "s_barrier", "buffer_", "flat_", "s_sleep", "scratch_")):
            if prev and prev[0] == op:
                prev[1] += 1
                prev[2] = max(prev[2], run_max)
                prev[3] += n_valu
            else:
                if prev:
                    print(f"{prev[4]:5d} {prev[0]:24s} x{prev[1]:3d}  maxv(since prev) {prev[2]:4d}  valu_before {prev[3]}")
                prev = [op, 1, run_max, n_valu, i]
            run_max = -1
            n_valu = 0
    if prev:
        print(f"{prev[4]:5d} {prev[0]:24s} x{prev[1]:3d}  maxv(since prev) {prev[2]:4d}  valu_before {prev[3]}")
    break
