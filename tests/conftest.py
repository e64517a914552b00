import contextlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--rehearse-on-model", action="store_true", default=False,
                     help="run the `-m gpu` tests' CODE on the CPU against the kernels' functional model (tests/wavesim): host tensors, "
                          "synchronous 'streams'; full-size and subprocess cases are skipped.  A rehearsal of the GPU suite in the GPU-less "
                          "authoring container -- it is not the GPU run and proves nothing about the hardware.")
    parser.addoption("--rehearse-on-code-object", action="store_true", default=False,
                     help="like --rehearse-on-model, but every kernel launch is executed from the BUILT gfx950 code objects by the instruction-level "
                          "interpreter (tests/gfx950_exec.py); the full-size configurations are skipped (hours).  A one-off campaign "
                          "(tools/README.md), not part of the CPU suite.")


def pytest_configure(config):
    if config.getoption("--rehearse-on-code-object"):
        config.option.rehearse_on_model = True
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hardware_only: a gpu test that cannot be rehearsed on the functional model (size, subprocess, real runtime)")
    if config.getoption("--rehearse-on-model"):
        _enter_rehearsal()
    if config.getoption("--rehearse-on-code-object"):
        _enter_code_object_rehearsal()


def pytest_sessionstart(session):
    """CPU runs: build the kernels' functional model and its variants (spin limit 0, AddressSanitizer, LDS access profile) up
    front and in parallel instead of one after the other inside the first tests that need them."""
    config = session.config
    if config.getoption("markexpr", "") == "gpu" and not config.getoption("--rehearse-on-model"):
        return  # the hardware run does not touch the model
    from concurrent.futures import ThreadPoolExecutor

    from ndzip_amd import build as hipbuild
    from tests.wavesim import build as simbuild

    # A fresh checkout: cross-compile the product library the C-ABI / adaptor / CLI tests load (what __graft_entry__.build()
    # does) and the checkers (the C oracle and, where /root/reference exists, the real serial reference, oracle/_ref).  A build
    # that fails ends the session here, with the compiler's message -- not later, as a missing library inside some test.
    try:
        hipbuild.build()
        hipbuild.build_test_variants()
    except Exception as e:
        pytest.exit(f"product build failed:\n{e}", returncode=2)
    try:
        from oracle import oracle

        oracle.build(ref=True)
    except Exception as e:
        pytest.exit(f"oracle build failed:\n{e}", returncode=2)

    jobs = [dict(), dict(variant="spin0", defines=("NDZIP_LOOKBACK_SPIN_LIMIT=0",)), dict(variant="asan", extra_flags=simbuild.ASAN_FLAGS),
            dict(variant="ldsprof", defines=("WAVESIM_LDSPROF",), kernel_flags=simbuild.LDSPROF_FLAGS)]
    try:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(lambda kw: simbuild.build(**kw), jobs))
    except Exception as e:
        pytest.exit(f"functional-model build failed:\n{e}", returncode=2)


def pytest_collection_modifyitems(config, items):
    # The driver runs `-m gpu -x`: the first failure ends the run.  Tests that could never be rehearsed on the functional model
    # (hardware_only: tools linked against the real runtime, the memory-mapped CLI path, multi-process runs) go to the END of the
    # session, so that a surprise in one of them cannot hide the parity results of everything that was rehearsed.
    # Inside the rehearsed part the same idea, by what has been on silicon before: the 32-bit profiles' parity tests first (their
    # kernels descend from the ones round 1 ran on an MI355X), then everything that touches a 64-bit profile (three rounds of
    # hand-written DPP sequences that never ran) and the tests that mix both, then the mechanisms that never ran at all (hipGraph
    # capture, background load, the file tools) -- so that a failure in a later class still leaves the earlier classes' results.
    mixed = ("test_known_answers_from_reference", "test_smoke_entry_point", "test_sharded_codec_single_rank", "test_full_size_configs")
    late_files = ("test_hip_graph", "test_hip_stress", "test_hip_cli", "test_hip_sharded_mp", "test_hip_sharded_rccl", "test_cpp_adaptor")

    def risk(item):
        nid = item.nodeid.lower()
        if "hardware_only" in item.keywords:
            return 3
        if any(f in nid for f in late_files):
            return 2
        if "test_zz_both_f64_decoder_kernels_agreed" in nid:
            return 1.5  # (behind every test that may have recorded a disagreement)
        if any(k in nid for k in ("float64", "f64", "double")) or any(m in nid for m in mixed):
            return 1
        return 0

    items.sort(key=risk)  # (stable: the order inside each class stays)
    if not config.getoption("--rehearse-on-model"):
        return
    skip = pytest.mark.skip(reason="hardware only: not part of the rehearsal on the functional model")
    too_long = pytest.mark.skip(reason="full-size configuration / stress case: hours on the instruction-level interpreter")
    for item in items:
        if "hardware_only" in item.keywords:
            item.add_marker(skip)
        elif config.getoption("--rehearse-on-code-object") and ("full_size" in item.name or "test_hip_stress" in item.nodeid):
            item.add_marker(too_long)  # (the stress cases: tens of thousands of hypercubes)


_rehearsal = contextlib.ExitStack()


def _enter_rehearsal():
    """Host tensors for device tensors, the model library for the HIP library, no-op streams for torch.cuda's."""
    import torch

    from tests import util
    from tests.wavesim import sim

    _rehearsal.enter_context(sim.active(cus=3, blocks_per_cu=2))
    util.DEVICE = "cpu"

    class _Stream:
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def synchronize(self):
            pass

    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream = _Stream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.is_available = lambda: True


def _enter_code_object_rehearsal():
    """On top of the model rehearsal: every kernel launch of the model's host side runs as gfx950 code (tests/gfx950_exec.py)."""
    import tempfile

    from ndzip_amd import hip
    from tests import gfx950_exec as gx
    from tests.wavesim import build as simbuild

    bridge = gx.Bridge(simbuild.build(), [hip.LIB_PATH, hip.STAGES_LIB_PATH], tempfile.mkdtemp(prefix="gfx950_rehearsal"))
    _rehearsal.enter_context(bridge)
    _code_object_bridge.append(bridge)


_code_object_bridge = []


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    if _code_object_bridge and outcome.excinfo is None:
        b = _code_object_bridge[0]
        if b.error is not None:  # (an interpreter error cannot cross the model's C frames: it is kept and the launch falls back)
            err, b.error = b.error, None
            raise err


def pytest_sessionfinish(session, exitstatus):
    if _code_object_bridge:  # one line per process (xdist worker) with what the interpreter did: tools/README.md sums them up
        from tests import gfx950_exec as gx

        b = _code_object_bridge[0]
        with open(os.path.join(os.environ.get("CODE_OBJECT_REHEARSAL_STATS", "/tmp"), f"code_object_rehearsal.{os.getpid()}.txt"), "w") as f:
            f.write(f"launches {b.total_launches} instructions {b.total_instructions} hazards {len(gx.HAZARD_LOG)} waits {len(gx.WAIT_LOG)}\n")


@pytest.fixture(scope="session")
def rehearsal(request):
    return request.config.getoption("--rehearse-on-model")


@pytest.fixture(scope="session")
def hiplib(rehearsal):
    """The in-tree HIP library; GPU tests fail loudly (no skip, no fallback) if it is missing."""
    from ndzip_amd import hip

    return hip.lib()  # (in a rehearsal hip._lib already is the model, see _enter_rehearsal)


@pytest.fixture(scope="session")
def cuda_device(rehearsal):
    import torch

    if rehearsal:
        return torch.device("cpu")
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")
