import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running test")


@pytest.fixture(scope="session", autouse=True)
def _native_host_lib():
    """The C++ host runtime is plain g++ code: build it on demand for CPU tests."""
    from d4pg_pytorch_amd import build
    if os.path.exists(os.path.join(build.CSRC, build.HOST_SOURCES[0])):
        build.build_host()
    yield


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
