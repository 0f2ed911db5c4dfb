import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        # A GPU test takes seconds (the whole -m gpu suite ~95 s, plus 1-2 min of first `import torch` on a cold box). One that runs
        # for minutes is stuck — a stream waiting for an event nobody records, a kernel that does not drain: with pytest-timeout
        # (in this image) it then fails with the Python stacks of all threads in the log and the process EXITS, instead of
        # sitting silent until the box's watchdog kills the run.
        if config.pluginmanager.hasplugin("timeout"):
            for item in items:
                if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                    item.add_marker(pytest.mark.timeout(300, method="thread"))
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
