#!/bin/bash
# usage: tools/pmc.sh <outfile> [bench args]; collects several PMC passes for the ndzip kernels and summarises them
OUT=$1; shift
export TMPDIR=/tmp; R=$PWD; P=/tmp/pmc_$$; mkdir -p $P
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o stats -- $B > $P/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $P/pmc1 -o pmc1 -- $B > $P/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $P/pmc2 -o pmc2 -- $B > $P/pmc2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d $P/pmc3 -o pmc3 -- $B > $P/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $P/pmc4 -o pmc4 -- $B > $P/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_STALL_sum --output-format csv -d $P/pmc5 -o pmc5 -- $B > $P/pmc5.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d $P/pmc6 -o pmc6 -- $B > $P/pmc6.log 2>&1
cd $R
# TRAFFIC_KEY (e.g. float32-512x512x512) also merges the FETCH/WRITE bytes of this run into gpurun_out/traffic.json
if [ -n "$TRAFFIC_KEY" ]; then
  python tools/prof_summary.py $P --traffic "$TRAFFIC_KEY" gpurun_out/traffic.json "$OUT (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" > $OUT 2>&1
else
  python tools/prof_summary.py $P > $OUT 2>&1
fi
rm -rf $P
