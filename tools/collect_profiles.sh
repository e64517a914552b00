#!/bin/bash
# After a gpurun call of tools/gpu_round_batch.sh / tools/_call1.sh <tag>: copy what the judge should see from the scratch
# directory gpurun_out/ into profiles/ (tracked) and regenerate profiles/traffic.json from the merged counters.
# usage: tools/collect_profiles.sh <tag-in-gpurun_out> <name-in-profiles>      e.g. tools/collect_profiles.sh r03a r03
set -e
src=${1:?tag}; dst=${2:-$1}
cd "$(dirname "$0")/.."
for f in smoke gputest bench_n1.json configs workgroups_per_cu rocprofv3_summary rocprofv3_summary_f64_2d ab_variants ab_variants_cfg1 ab_variants_f64_2d ab_variants_f64_3d phase_timing two_process_stress rocminfo; do
  for ext in "" .txt; do
    [ -f "gpurun_out/${src}_$f$ext" ] && cp "gpurun_out/${src}_$f$ext" "profiles/${dst}_$f$ext"
  done
done
[ -f gpurun_out/traffic.json ] && cp gpurun_out/traffic.json profiles/traffic.json
ls -la profiles/${dst}_* 2>/dev/null
