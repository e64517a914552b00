#!/bin/bash
# Background poller for a gated GPU: every $2 seconds try ONE gpurun call of the given batch; a refused call (rc 2/3) costs nothing.
# A poll is skipped while ndzip_amd/csrc, bench.py or tests/ have uncommitted edits or the library is older than its sources, so the
# snapshot that reaches the box is always a committed, built state. Stops after the first call that actually ran.
# usage: tools/gpu_poller.sh <tag> [interval_s]
tag=${1:?tag}; iv=${2:-600}
cd "$(dirname "$0")/.."
while true; do
  if [ -n "$(git status --porcelain ndzip_amd bench.py tests include __graft_entry__.py oracle tools/gpu_r04_first.sh)" ]; then
    echo "$(date +%T) skip: uncommitted edits"; sleep 120; continue
  fi
  lib=ndzip_amd/libndzip_hip.so
  if [ ! -f $lib ] || [ -n "$(find ndzip_amd/csrc -newer $lib -type f \( -name '*.hip' -o -name '*.hpp' -o -name '*.inl' \) | head -1)" ]; then
    echo "$(date +%T) skip: library stale"; sleep 120; continue
  fi
  /usr/local/graft/bin/gpurun --timeout 3400 -- "bash tools/gpu_r04_first.sh $tag" > /tmp/gpurun_$tag.log 2>&1
  rc=$?
  echo "$(date +%T) rc=$rc $(git rev-parse --short HEAD)"
  if [ $rc -ne 2 ] && [ $rc -ne 3 ]; then echo "RAN at $(git rev-parse --short HEAD)"; exit 0; fi
  sleep $iv
done
