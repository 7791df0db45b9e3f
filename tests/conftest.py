import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests on a box without a device: skipped, not run (the product exits loudly without its GPU --
    # INTEGRATION.md -- which would take pytest down with it).  /dev/kfd is the ROCm compute device node.
    if os.path.exists("/dev/kfd") or os.environ.get("WTAMD_TESTS_ASSUME_GPU"):
        return
    skip = pytest.mark.skip(reason="no ROCm device on this box (/dev/kfd missing)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
