#!/bin/bash
# Background poller (costs no builder turns): every INTERVAL seconds, if no kernel / build source has uncommitted edits, try the
# evidence batch once. A refused call returns at once and charges nothing; the loop ends at the first call that actually ran.
# usage: tools/gpu_poll.sh <tag> [interval-seconds] [batch-script]
tag=${1:-r04a}; interval=${2:-900}; batch=${3:-tools/gpu_r04_first.sh}
cd "$(dirname "$0")/.."
while true; do
  if [ -z "$(git status --porcelain -- ndzip_amd include bench.py __graft_entry__.py oracle tests/util.py)" ]; then
    python -c "import __graft_entry__ as g; g.build()" > /tmp/gpu_poll_build.log 2>&1 || { sleep $interval; continue; }
    /usr/local/graft/bin/gpurun --timeout 3000 -- "bash $batch $tag" > /tmp/gpu_poll_last.log 2>&1
    if ! grep -q "status=refused\|status=unavailable\|rc=None" /tmp/gpu_poll_last.log; then
      date >> /tmp/gpu_poll_last.log; echo "GPU CALL RAN"; tail -60 /tmp/gpu_poll_last.log; exit 0
    fi
  fi
  sleep $interval
done
