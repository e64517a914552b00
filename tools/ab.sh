#!/bin/bash
# A/B a set of library variants (ndzip_amd/_variants/*.so, built by tools/build_variant.sh) in one session: compress-only
# timing, interleaved rounds.  usage: tools/ab.sh "<variant names>" [bench args]      ("main" = the in-tree library)
V="$1"; shift
for round in 1 2; do
  for v in $V; do
    echo -n "round $round $v: "
    lib=$PWD/ndzip_amd/_variants/$v.so; [ "$v" = main ] && lib=$PWD/ndzip_amd/libndzip_hip.so
    python bench.py --lib "$lib" --steps 20 --warmup 3 --no-cpu-baseline --compress-only "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"
  done
done
