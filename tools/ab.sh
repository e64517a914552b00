#!/bin/bash
# A/B a set of library variants (ndzip_amd/_variants/*.so, built by tools/build_variant.sh) in one session, interleaved rounds.
# usage: [AB_MODE=both] tools/ab.sh "<variant names>" [bench args]      ("main" = the in-tree library)
# AB_MODE=compress (default): compress-only timing; both: compress and decompress launch times of a full bench step
V="$1"; shift
MODE=${AB_MODE:-compress}
for round in 1 2; do
  for v in $V; do
    echo -n "round $round $v: "
    lib=$PWD/ndzip_amd/_variants/$v.so; [ "$v" = main ] && lib=$PWD/ndzip_amd/libndzip_hip.so
    if [ "$MODE" = both ]; then
      python bench.py --lib "$lib" --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('compress_ms', r['launch_ms'], 'frac', r['frac'], '| decompress_ms', r['decompress']['launch_ms'], 'frac', r['decompress']['frac'])"
    else
      python bench.py --lib "$lib" --steps 20 --warmup 3 --no-cpu-baseline --compress-only "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"
    fi
  done
done
