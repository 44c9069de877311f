import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import t4oracle
    t4oracle.lib()
    return t4oracle


@pytest.fixture(scope="session")
def t4k():
    """The product library, initialised on cuda:0.  Fails loudly (no fallback)."""
    from tensorforth_amd.lib import load
    h = load(os.environ.get("T4K_LIB") or None)          # T4K_LIB: the LAB library (tests/test_gpu_switches.py lab matrix)
    h.init(0)
    return h
