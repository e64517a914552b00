tag=r03a
mkdir -p gpurun_out
O=gpurun_out/$tag
rocminfo | grep -E "gfx|Compute Unit" | head -4 > ${O}_rocminfo.txt
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -60) > ${O}_gputest.txt
(timeout 180 python __graft_entry__.py smoke 2>&1 | tail -3) >> ${O}_gputest.txt
timeout 400 python bench.py > ${O}_bench_n1.json 2> ${O}_bench_n1.err
(timeout 900 bash tools/bench_configs.sh 2>&1) > ${O}_configs.txt
TRAFFIC_KEY=float32-512x512x512 timeout 900 bash tools/pmc.sh ${O}_rocprofv3_summary.txt
tail -8 ${O}_gputest.txt; cat ${O}_bench_n1.json; cat ${O}_configs.txt
