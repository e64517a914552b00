#!/bin/bash
# A/B a set of library variants (ndzip_amd/_variants/*.so) in one session: compress-only timing, interleaved rounds
# usage: tools/ab.sh "<variant names>" [bench args]
V="$1"; shift
for round in 1 2; do
  for v in $V; do
    echo -n "round $round $v: "
    NDZIP_HIP_LIB=$PWD/ndzip_amd/_variants/$v.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --compress-only "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'])"
  done
done
