#!/bin/bash
# Short (about 12 minutes) version of tools/gpu_round_batch.sh, highest-value evidence first; every step writes its own file under
# gpurun_out/<tag>_* as soon as it ends, so a call that is cut off still leaves what finished.
# usage: tools/gpu_quick_batch.sh <tag>
tag=${1:-rNN}
mkdir -p gpurun_out
O=gpurun_out/$tag
rocminfo | grep -E "gfx|Compute Unit" | head -4 > ${O}_rocminfo.txt
(timeout 180 python __graft_entry__.py smoke 2>&1 | tail -3) > ${O}_smoke.txt
timeout 300 python bench.py > ${O}_bench_n1.json 2> ${O}_bench_n1.err
TRAFFIC_KEY=float32-512x512x512 timeout 420 bash tools/pmc.sh ${O}_rocprofv3_summary.txt
(timeout 420 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > ${O}_gputest.txt
(timeout 300 bash tools/bench_configs.sh 2>&1) > ${O}_configs.txt
cat ${O}_smoke.txt; tail -5 ${O}_gputest.txt; cat ${O}_bench_n1.json; tail -30 ${O}_configs.txt
