import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library's default arithmetic is the range-guarded split-f16 mode ("f16x3", sola_amd/module.py).  The parity tests state the
# mode they test; modules they build without saying so are the exact-f32 baseline.  Entry-point tests (subprocesses) drop this
# variable again and run the real default.
os.environ.setdefault("SOLA_PRECISION", "f32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def split_kernels_at_every_size():
    """Round 4: a precision-"f16x3" inference call over at most 4096 object-token rows (one sample per call) runs the exact-f32 kernels -
    faster there, and exact (sola_tune "infer_f32_rows").  The parity tests of the split-f16 path use small shapes on purpose; they keep
    testing the split kernels (0 = no routing).  tests/test_gpu_fast.py::test_few_row_calls_of_the_default_mode_run_exact_f32 covers the
    routing itself; entry-point tests (subprocesses) run the real default."""
    from sola_amd import _lib
    try:
        handle = _lib.lib()
    except _lib.SolaLibraryError:
        handle = None  # CPU box without a built library: the -m "not gpu" tests that need it say so themselves
    if handle is not None:
        _lib.check(handle.sola_tune(b"infer_f32_rows", 0), "tune infer_f32_rows")  # a rejected key is an error, not something to swallow
    yield


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def small_golden():
    return _load("small_golden.npz")


@pytest.fixture(scope="session")
def full_golden():
    return _load("full_golden.npz")


@pytest.fixture(scope="session")
def iou_golden():
    return _load("iou_golden.npz")


def case_dict(golden, ci):
    pre = f"c{ci}."
    return {k[len(pre):]: golden[k] for k in golden.files if k.startswith(pre)}
