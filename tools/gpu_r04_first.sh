#!/bin/bash
# Round-4 first evidence call: smoke, default bench line, the whole -m gpu suite (no -x: every failure listed), rocprofv3 kernel trace
# + PMC passes (writes gpurun_out/traffic.json), then every BASELINE config. One output file per step.
# usage: tools/gpu_r04_first.sh <tag>
tag=${1:-r04a}
mkdir -p gpurun_out
O=gpurun_out/$tag
rocminfo | grep -E "gfx|Compute Unit" | head -4 > ${O}_rocminfo.txt
(timeout 240 python __graft_entry__.py smoke 2>&1 | tail -5) > ${O}_smoke.txt
timeout 400 python bench.py > ${O}_bench_n1.json 2> ${O}_bench_n1.err
(timeout 1200 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -60) > ${O}_gputest.txt
TRAFFIC_KEY=float32-512x512x512 timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary.txt
(timeout 400 bash tools/bench_configs.sh 2>&1) > ${O}_configs.txt
cat ${O}_smoke.txt; tail -15 ${O}_gputest.txt; cat ${O}_bench_n1.json; tail -5 ${O}_bench_n1.err; tail -30 ${O}_configs.txt
