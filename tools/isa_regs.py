#!/usr/bin/env python
"""VGPR / scratch / occupancy per kernel from a hipcc -S file.  usage: isa_regs.py file.s [substring]"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n.*?s_endpgm.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", s, re.S | re.M):
    name = m.group(1)
    if key not in name:
        continue
    short = re.sub(r"^_ZN9ndzip_hip12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"EEEv.*", "", short)
    print(f"{short:40s} vgprs {m.group(2):>4s} scratch {m.group(3):>4s} occupancy {m.group(4)}")
