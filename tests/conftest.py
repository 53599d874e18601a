"""pytest wiring: `gpu` marker, import paths, golden-fixture loader."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rule-guided-music_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly -- never skip silently -- when SELECTED (-m gpu) without a GPU; a plain `pytest tests` on a CPU-only
    box skips them instead of reporting hundreds of errors."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a MI355X: run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(params=["fp32", "bf16x3", "bf16x3_presplit"])
def precision(request):
    """Run a GPU parity test under both GEMM arithmetics (exact fp32 MFMA and the bf16x3 split)."""
    from rgm import native as R
    R.set_gemm_precision(request.param)
    yield request.param
    R.set_gemm_precision("fp32")


def chord_test_roll(seed):
    """The input of tests/golden/chord_quantise.npz, rebuilt from its seed (make_golden.py chord_test_roll)."""
    rng = np.random.RandomState(seed)
    roll = (rng.rand(3, 3, 128, 256).astype(np.float32) * 2.4 - 1.2)
    roll[rng.rand(*roll.shape) < 0.3] = -0.95
    roll[rng.rand(*roll.shape) < 0.1] = np.float32(-0.9500001)
    return roll
