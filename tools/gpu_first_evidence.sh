#!/bin/bash
# First GPU call of a round: the full parity suite, smoke(), one bench line and a kernel trace of the same bench command.
# Everything lands in gpurun_out/ (copied to profiles/ by hand once read).
tag=${1:-r02}
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/${tag}_rocminfo.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/${tag}_gputest.txt
(timeout 180 python __graft_entry__.py smoke 2>&1 | tail -3) >> gpurun_out/${tag}_gputest.txt
timeout 400 python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
cat gpurun_out/${tag}_gputest.txt gpurun_out/${tag}_bench_n1.json
tail -3 gpurun_out/${tag}_bench_n1.err
