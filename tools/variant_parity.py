"""Parity of a library VARIANT against the oracle on the GPU: `python tools/variant_parity.py ndzip_amd/_variants/plain.so`.
The bisecting companion of the -m gpu suite: `plain.so` is the product source without the EXEC-masked assembly and without the
v_readfirstlane pins (gfx950_lds.hpp), built by ndzip_amd/build.py::build_test_variants.  Prints one line per case and a verdict;
exit status 0 iff every stream and round trip is bit-exact."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndzip_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.abspath(sys.argv[1])
from ndzip_amd.synth import synth_numpy  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.util import device_compress, device_decompress, same_bits  # noqa: E402

CASES = [(np.float32, (4096 * 37 + 11,)), (np.float32, (200, 330)), (np.float32, (70, 50, 36)), (np.float32, (64, 64, 128)),
         (np.float64, (4096 * 9 + 5,)), (np.float64, (130, 200)), (np.float64, (40, 48, 33)), (np.float32, (256, 256, 256)),
         (np.float64, (2048, 2048))]
bad = 0
for dtype, shape in CASES:
    for noise in (0xFF, 0xFFFFFF, 0):
        data = synth_numpy(shape, dtype, seed=5, noise_mask=noise)
        want = oracle.compress(data)
        try:
            got = device_compress(data)
            ok_c = got.shape == want.shape and np.array_equal(got, want)
        except Exception as e:  # (a device error word, a fault: one line, next case)
            ok_c = False
            print(f"  compress raised {type(e).__name__}: {str(e)[:160]}")
        # each 64-bit decoder kernel on its own (the default call of device_decompress runs both and asserts they agree)
        ok_d, which = True, ""
        for work_items in ((0,) if dtype == np.float32 else (128, 256)):
            try:
                ok = same_bits(device_decompress(want, dtype, shape, f64_work_items=work_items), data)
            except Exception as e:
                ok = False
                print(f"  decompress raised {type(e).__name__}: {str(e)[:160]}")
            if not ok:
                ok_d, which = False, which + f" [{work_items or 128} work-items]"
        bad += (not ok_c) + (not ok_d)
        print(f"{np.dtype(dtype).name} {shape} noise {noise:#x}: compress {'ok' if ok_c else 'DIFFERS'}, decompress {'ok' if ok_d else 'DIFFERS' + which}", flush=True)
print("VARIANT", os.path.basename(sys.argv[1]), "BIT-EXACT" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
