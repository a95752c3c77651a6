import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must not silently pass on a box without a GPU: they fail loudly when selected."""
    import torch
    if torch.cuda.is_available():
        return
    selected_gpu = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if selected_gpu:
        return  # let them run and fail: the product path has no fallback
    skip = pytest.mark.skip(reason="no GPU in this container (selected without -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
