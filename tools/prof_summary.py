#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace / counter collection) for the ndzip kernels only.

usage: prof_summary.py <rocprof output dir> [name filter, default 'ndzip_hip']
Prints, per kernel: dispatch count, average duration (kernel trace) and average value per dispatch of every
counter (counter collection).  The raw CSVs hold thousands of torch dispatches from the input generator and
are left on the GPU box; only this summary is committed under profiles/.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(decompress_kernel_wide|decompress_kernel|compress_kernel_wide|compress_kernel_db|compress_kernel|border_kernel|debug_\w+|offset_header\w*|store_length\w*)<?([^>(]*)", name)
    if m:
        return (m.group(1) + "<" + m.group(2) + ">").replace("ndzip_hip::", "")
    return name[:80]


def traffic_entry(counters, source):
    """HBM bytes per launch of the compress and decompress kernels from the FETCH_SIZE / WRITE_SIZE passes (KiB units; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ndzip_amd.build import kernels_fingerprint

    entry, raw = {"source": source, "kernels": kernels_fingerprint()}, {}
    for kernel, which in (("compress_kernel", "compress"), ("decompress_kernel", "decompress")):
        hits = [v for k, v in counters.items() if k.startswith(kernel) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if not hits:
            continue
        c = max(hits, key=lambda v: v["FETCH_SIZE"] + v["WRITE_SIZE"])  # (the codec kernel, not a stage or border kernel)
        raw[which] = {"FETCH_SIZE_KiB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(c["WRITE_SIZE"], 1)}
        entry[which + "_hbm_bytes_per_launch"] = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
    entry["raw"] = raw
    return entry


def main():
    # prof_summary.py <dir> [filter] [--traffic <key> <traffic.json> <source note>]: also merge the HBM bytes into traffic.json
    traffic = None
    if "--traffic" in sys.argv:
        i = sys.argv.index("--traffic")
        traffic = sys.argv[i + 1: i + 4]
        del sys.argv[i: i + 4]
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "ndzip_hip"
    averages = defaultdict(dict)
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        dur = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                if filt in name:
                    dur[short(name)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print(f"# {os.path.relpath(path, root)}")
        print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
        for k, v in sorted(dur.items()):
            print(f"{k:60s} {len(v):6d} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:10.2f} {max(v) / 1e3:10.2f}")
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                if filt in name:
                    k = short(name)
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                               row.get("Workgroup_Size"), row.get("Grid_Size"))
        print(f"# {os.path.relpath(path, root)}")
        for k in sorted(acc):
            print(f"{k}  vgpr/agpr/sgpr/lds/wg/grid={meta[k]}")
            for c, v in sorted(acc[k].items()):
                print(f"    {c:28s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
                averages[k][c] = sum(v) / len(v)
    # derived: how many wavefronts were resident per CU on average (the census the persistent grid's sizing rests on: 16 = four
    # 256-thread workgroups per CU), waiting and VALU shares of the wave time.  SQ_* wave counters tick in quad-cycles
    # (MI355X guide), GRBM_GUI_ACTIVE in cycles; both are per launch here.
    print("# derived")
    for k, c in sorted(averages.items()):
        if "SQ_WAVE_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            cus = float(os.environ.get("NDZIP_PROF_CUS", "256"))
            print(f"{k}: average resident wavefronts per CU = {4 * c['SQ_WAVE_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * cus):.1f} (of {cus:.0f} CUs)")
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
            print(f"{k}: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}"
                  + (f", SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.2f}" if "SQ_ACTIVE_INST_VALU" in c else ""))
        if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
            print(f"{k}: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.2f}")
    if traffic:
        import json

        key, path, source = traffic
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[key] = traffic_entry(averages, source)
        with open(path, "w") as f:
            json.dump(data, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
