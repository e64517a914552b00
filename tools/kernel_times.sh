#!/bin/bash
# per-kernel average durations of one bench configuration: tools/kernel_times.sh [bench args]
export TMPDIR=/tmp; R=$PWD; P=/tmp/kt_$$
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o s -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $P.log 2>&1
python3 - $P <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(1)
for r in csv.DictReader(open(fs[0])):
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
PY
rm -rf $P $P.log
