#!/bin/bash
# The kernels' C++ on the functional model built with UndefinedBehaviorSanitizer (every finding aborts): the model's own suites and the
# rehearsed `-m gpu` suite.  CPU only, ~6 min.  Counterpart of tools/asan_rehearsal.sh.     usage: tools/ubsan_rehearsal.sh [pytest args]
cd "$(dirname "$0")/.."
python -c "from tests.wavesim import build as b; b.build(variant='ubsan', extra_flags=b.UBSAN_FLAGS)" || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 WAVESIM_VARIANT=ubsan \
  python -m pytest tests/test_wavesim_codec.py tests/test_wavesim_stages.py tests/test_wavesim_fuzz.py -q -x -p no:cacheprovider "$@" && \
LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 WAVESIM_VARIANT=ubsan \
  python -m pytest tests -m gpu --rehearse-on-model -x -q -p no:cacheprovider "$@"
