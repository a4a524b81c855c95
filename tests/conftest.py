import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_c():
    """The plain-C oracle (oracle/hotpath_c.c), built on demand with gcc."""
    import subprocess

    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import hotpath_c

    return hotpath_c
