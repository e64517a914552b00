#!/usr/bin/env python
"""Instruction-mix histogram of one kernel from a hipcc -save-temps .s file.
usage: isa_mix.py file.s <substring of mangled kernel name> [topN]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if key not in name:
        continue
    c = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line[0] in ".;/" or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    print(name[:90], "total", sum(c.values()))
    print(", ".join(f"{k}:{v}" for k, v in c.most_common(top)))
