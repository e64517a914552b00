#!/usr/bin/env python
"""Dump per-dispatch counters of a rocprofv3 --pmc CSV run: pmc_dump.py <dir> [kernel name filter]"""
import collections, csv, glob, sys
rows = collections.defaultdict(dict)
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            rows[(int(r["Dispatch_Id"]), r["Kernel_Name"][:40])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    print(k, rows[k])
