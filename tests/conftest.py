import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests` in the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def g_ops():
    return _load("ops.npz")


@pytest.fixture(scope="session")
def g_minkunet():
    return _load("model_minkunet.npz")


@pytest.fixture(scope="session")
def g_miou():
    return _load("miou_minkunet.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms():
    return _load("model_minkunet_ms.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms_mm():
    return _load("model_minkunet_ms_mm.npz")


@pytest.fixture(scope="session")
def g_minkunet_ms_kd():
    return _load("model_minkunet_ms_kd.npz")


@pytest.fixture(scope="session")
def g_multiscan():
    return _load("multiscan.npz")
