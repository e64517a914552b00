#!/bin/bash
# timing experiments on the GPU box (compress only) on the variant built by
#   tools/build_variant.sh knobs --lab -DNDZIP_EXP_KNOBS -DNDZIP_EXP_ABLATION
# (the production library reads no knobs):
#   NDZIP_HIP_EXP bit0 = no look-back (fake offsets), bit1 = no copy-out, bit2 = no plane writes
#   NDZIP_HIP_BPC = cap on resident workgroups per CU
LIB=${LIB:-$PWD/ndzip_amd/_variants/knobs.so}
run() { echo -n "EXP=$1 BPC=$2: "; NDZIP_HIP_EXP=$1 NDZIP_HIP_BPC=$2 python bench.py --lib "$LIB" --steps 20 --warmup 3 --no-cpu-baseline --compress-only "${@:3}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compress_ms', d['roofline']['launch_ms'])"; }
for e in ${EXPS:-0 1 2 4 3 7}; do run $e 0 "$@"; done
for b in ${BPCS:-1 2}; do run 0 $b "$@"; run 1 $b "$@"; done
