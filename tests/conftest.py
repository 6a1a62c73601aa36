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
