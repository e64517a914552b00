#!/bin/bash
export TMPDIR=/tmp; R=$PWD
for e in 0 2 4 6; do
  P=/tmp/pl_$e; rm -rf $P; mkdir -p $P; cd /tmp
  NDZIP_HIP_EXP=$e rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS --output-format csv -d $P -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --compress-only "$@" > $P/log 2>&1
  cd $R; echo "== EXP=$e"; python tools/prof_summary.py $P | grep -E "compress|SQ_"
done
