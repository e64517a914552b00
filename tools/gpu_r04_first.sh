#!/bin/bash
# Round-4 first evidence call: smoke, the whole -m gpu suite, PMC passes (traffic.json) BEFORE the bench line, all configs.
# Every step has its own timeout and writes its own file under gpurun_out/<tag>_*.
tag=${1:-r04a}
mkdir -p gpurun_out
O=gpurun_out/$tag
rocminfo | grep -E "gfx|Compute Unit" | head -4 > ${O}_rocminfo.txt
(timeout 240 python __graft_entry__.py smoke 2>&1 | tail -5) > ${O}_smoke.txt
cat ${O}_smoke.txt
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --timeout 180 2>&1 | tail -80) > ${O}_gputest.txt
tail -30 ${O}_gputest.txt
TRAFFIC_KEY=float32-512x512x512 timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary.txt
cp gpurun_out/traffic.json profiles/traffic.json 2>/dev/null
timeout 400 python bench.py > ${O}_bench_n1.json 2> ${O}_bench_n1.err
cat ${O}_bench_n1.json; tail -5 ${O}_bench_n1.err
(timeout 700 bash tools/bench_configs.sh 2>&1) > ${O}_configs.txt
cat ${O}_configs.txt
