#!/bin/bash
# Rebuild every A/B variant of `tools/gpu_batch.sh explain` against the CURRENT sources (CPU; after any change under ndzip_amd/csrc) and
# check the lab ones bit-exact on the instruction-level interpreter.  History variants (r01 .. r04, r05a) are built from their commits
# once and kept; pass --history to rebuild r04 / r05a too.        usage: tools/rebuild_variants.sh [--history]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--history" ]; then
  bash tools/build_history_variant.sh r04 e0ea2d5 -mllvm -amdgpu-atomic-optimizer-strategy=None     # end of round 4
  # the tree before round 5's change of the post-B3 order (parent of the commit "the transposes run while the copy-out's stores ...")
  C=$(git log --format=%H --grep="the transposes run while the copy-out" | tail -1)
  bash tools/build_history_variant.sh r05a "$(git rev-parse "$C~1")" -mllvm -amdgpu-atomic-optimizer-strategy=None
fi
while read -r name flags; do
  [ -z "$name" ] && continue
  # shellcheck disable=SC2086
  bash tools/build_variant.sh "$name" --lab $flags 2>&1 | grep -v "warning\|^In file\|tick_dummy\|^ *[0-9]* |\|\^" | tail -1
done <<'LIST'
winpub -DNDZIP_EXP_WINDOW_BEHIND_PUBLISH
trearly -DNDZIP_EXP_TRANSPOSE_BEFORE_LOOKBACK
cobatch2 -DNDZIP_EXP_COPYOUT_BATCH=2
wg3 -DNDZIP_EXP_DB_WAVES=3
plainloads -DNDZIP_PLAIN_INPUT_LOADS
timing -DNDZIP_EXP_KNOBS -DNDZIP_EXP_PHASE_TIMING
knobs -DNDZIP_EXP_KNOBS -DNDZIP_EXP_ABLATION
f64sched -DNDZIP_EXP_F64_NOCARRY
LIST
# HEAD compiled without the pass that (in this image's compiler) can sink a load past a barrier -- it sinks none in these kernels
# (tools/audit_machine_sink.py), so this is HEAD with 76-90 ALU instructions per kernel left where the scheduler put them: a bisecting aid
bash tools/build_variant.sh nosink -mllvm -disable-machine-sink 2>&1 | grep -v "warning\|^In file\|tick_dummy\|^ *[0-9]* |\|\^" | tail -1
python -c "from ndzip_amd import build; build.build_test_variants()"   # plain (no inline asm, no scalar pins), spin0
python tools/variant_parity_cpu.py winpub trearly cobatch2 wg3 plainloads f64sched nosink
