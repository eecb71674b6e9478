import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically where no device is visible so that a bare
    # `pytest tests/` works on the CPU container too.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_pending_deferred_reduce():
    """A test that fails between a deferred split-K conv and its consumer norm must not poison the next test."""
    yield
    ops = sys.modules.get("audioldm_with_lora_amd.ops")
    if ops is not None:
        ops.drop_pending()
