#!/bin/bash
# Round-4 second call (after tools/gpu_r04_first.sh has produced parity + bench + PMC): what explains the numbers.
#   A/B of HEAD against the libraries of the end of round 3 and round 2 and against HEAD with plain-policy input loads (cfg 2, cfg 1,
#   f64 2D / 3D compress + decompress), the workgroups-per-CU sweep, the phase timers of the f32 compress iteration, the PMC passes of
#   cfg 3 and of the f64 3D decoder in both mappings.  (The two-process stress is NOT part of this batch any more: it is the one
#   workload that has hung a box -- round 1 -- and a hung box is a strike; tools/gpu_two_process_stress.sh runs it on its own, last.)  Variants are built on the CPU beforehand
#   (tools/build_history_variant.sh r03 5bc12d1 ..., tools/build_variant.sh timing --lab ...).   usage: tools/gpu_r04_second.sh <tag>
tag=${1:-r04b}
mkdir -p gpurun_out
O=gpurun_out/$tag
V="main"; for v in r03 r02 plainloads; do [ -f ndzip_amd/_variants/$v.so ] && V="$V $v"; done
(timeout 700 bash tools/ab.sh "$V" 2>&1) > ${O}_ab_variants.txt; cat ${O}_ab_variants.txt
(timeout 400 bash tools/ab.sh "$V" --config 1 2>&1) > ${O}_ab_variants_cfg1.txt
for w in 0 3 2 1; do echo -n "workgroups per CU $w: "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --compress-only --workgroups-per-cu $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"; done > ${O}_workgroups_per_cu.txt 2>&1
cat ${O}_workgroups_per_cu.txt
(AB_MODE=both timeout 500 bash tools/ab.sh "$V" --config 3 2>&1) > ${O}_ab_variants_f64_2d.txt
(AB_MODE=both timeout 500 bash tools/ab.sh "$V" --shape 512,512,512 --dtype float64 2>&1) > ${O}_ab_variants_f64_3d.txt
cat ${O}_ab_variants_f64_2d.txt ${O}_ab_variants_f64_3d.txt
if [ -f ndzip_amd/_variants/timing.so ]; then
  (NDZIP_HIP_EXP=16 timeout 300 python bench.py --lib $PWD/ndzip_amd/_variants/timing.so --steps 3 --warmup 1 --no-cpu-baseline --compress-only 2>&1 | tail -40) > ${O}_phase_timing.txt
  tail -20 ${O}_phase_timing.txt
fi
TRAFFIC_KEY=float64-8192x8192 timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary_f64_2d.txt --config 3
timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary_f64_3d_decode_wide.txt --config 5
timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary_f64_3d_decode_128.txt --config 5 --f64-work-items 128
cp gpurun_out/traffic.json profiles/traffic.json 2>/dev/null
