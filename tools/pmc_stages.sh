#!/bin/bash
# LDS conflict cycles of the single-hypercube stage kernels (exact per-dispatch numbers)
export TMPDIR=/tmp; R=$PWD; P=/tmp/ps; rm -rf $P; mkdir -p $P; cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $P -o p -- python -m pytest $R/tests/test_hip_stages.py -q -x -p no:cacheprovider -k "$1" > $P/log 2>&1
cd $R; tail -2 $P/log
python - <<'PY'
import csv,glob,collections
rows=collections.defaultdict(list)
for f in glob.glob('/tmp/ps/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'debug_stage' in r['Kernel_Name']:
            rows[(r['Kernel_Name'][r['Kernel_Name'].find('debug_stage'):][:60], r['Dispatch_Id'])].append((r['Counter_Name'], float(r['Counter_Value'])))
for k in sorted(rows, key=lambda k:int(k[1])):
    print(k, dict(rows[k]))
PY
