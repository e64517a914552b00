#!/bin/bash
# One gpurun call's worth of evidence for a round (everything under gpurun_out/<tag>_*; copy what matters to profiles/):
#   1. the full -m gpu suite + smoke()                         (parity)
#   2. the default bench line                                   (the driver's command)
#   3. every BASELINE config + bookends, one line each          (tools/bench_configs.sh)
#   4. rocprofv3 kernel trace + PMC passes of the default bench (tools/pmc.sh)
#   5. A/B of the read-once input loads against plain loads     (tools/ab.sh; variant built on the CPU beforehand)
#   6. the two-processes-on-one-GPU stress of the sharded path, 6 x 40 iterations, each under its own timeout
# usage: tools/gpu_round_batch.sh <tag>     (about 25 minutes)
tag=${1:-rNN}
mkdir -p gpurun_out
O=gpurun_out/$tag
rocminfo | grep -E "gfx|Compute Unit" | head -4 > ${O}_rocminfo.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > ${O}_gputest.txt
(timeout 180 python __graft_entry__.py smoke 2>&1 | tail -3) >> ${O}_gputest.txt
timeout 400 python bench.py > ${O}_bench_n1.json 2> ${O}_bench_n1.err
(timeout 900 bash tools/bench_configs.sh 2>&1) > ${O}_configs.txt
TRAFFIC_KEY=float32-512x512x512 timeout 900 bash tools/pmc.sh ${O}_rocprofv3_summary.txt
TRAFFIC_KEY=float64-8192x8192 timeout 600 bash tools/pmc.sh ${O}_rocprofv3_summary_f64_2d.txt --config 3
# A/B of whatever variants were built on the CPU beforehand (tools/build_variant.sh, tools/build_history_variant.sh):
#   r03s1 = the library at the start of the second session of round 3 (before the bitop3 complement / decoder gather changes)
#   r02 = the round-2 library (3 workgroups per CU, branchy plane compaction, per-lane pointers), r01 = the round-1 pipeline
#   wg3 = HEAD held to 3 wavefronts per SIMD (f32 kernels): isolates what the 4th workgroup per CU buys
#   plainloads = HEAD without the nt input loads
V="main"; for v in wg3 r03s1 r02 plainloads r01; do [ -f ndzip_amd/_variants/$v.so ] && V="$V $v"; done
(timeout 900 bash tools/ab.sh "$V" 2>&1) > ${O}_ab_variants.txt
(timeout 600 bash tools/ab.sh "$V" --config 1 2>&1) > ${O}_ab_variants_cfg1.txt
# the same library at 4 / 3 / 2 workgroups per CU (ndzip_hip_compressor_set_max_workgroups_per_cu): what the occupancy alone buys
for w in 0 3 2; do echo -n "workgroups per CU $w: "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --compress-only --workgroups-per-cu $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"; done > ${O}_workgroups_per_cu.txt 2>&1
#   linear64 = 64-bit encoded runs linear in LDS (the round-1 layout) instead of XOR-swizzled (f64 only)
V64="main"; for v in r03s1 r02 linear64 plainloads r01; do [ -f ndzip_amd/_variants/$v.so ] && V64="$V64 $v"; done
(AB_MODE=both timeout 600 bash tools/ab.sh "$V64" --config 3 2>&1) > ${O}_ab_variants_f64_2d.txt
(AB_MODE=both timeout 600 bash tools/ab.sh "$V64" --shape 512,512,512 --dtype float64 2>&1) > ${O}_ab_variants_f64_3d.txt
# per-phase cycle totals of the f32 compress iteration (lab build with phase timers; NDZIP_HIP_EXP=16)
if [ -f ndzip_amd/_variants/timing.so ]; then
  (NDZIP_HIP_EXP=16 timeout 300 python bench.py --lib $PWD/ndzip_amd/_variants/timing.so --steps 3 --warmup 1 --no-cpu-baseline --compress-only 2>&1 | tail -40) > ${O}_phase_timing.txt
fi
for i in 1 2 3 4 5 6; do
  echo "== run $i" >> ${O}_two_process_stress.txt
  HSA_ENABLE_IPC_MODE_LEGACY=0 CHECK_EACH=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port $((29510 + i)) tools/sharded_stress.py 40 >> ${O}_two_process_stress.txt 2>&1
  echo "exit $?" >> ${O}_two_process_stress.txt
done
tail -5 ${O}_gputest.txt; cat ${O}_bench_n1.json; cat ${O}_configs.txt; cat ${O}_ab_variants.txt; tail -14 ${O}_two_process_stress.txt
