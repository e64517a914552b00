#!/bin/bash
# The rehearsed `-m gpu` suite (tests/conftest.py --rehearse-on-model: host tensors, the kernels' functional model) with the
# model built with AddressSanitizer: exact-size "device" buffers, LDS limited to what each launch asked for.  CPU only, ~4 min.
cd "$(dirname "$0")/.."
RT=$(python -c "from tests.wavesim import build as b; b.build(variant='asan', extra_flags=b.ASAN_FLAGS); print(b.asan_runtime())")
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 WAVESIM_VARIANT=asan \
  python -m pytest tests -m gpu --rehearse-on-model -x -q -p no:cacheprovider "$@"
