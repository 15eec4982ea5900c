import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle's C loops are test infrastructure; build them on demand (gcc is in the image)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle_loops.so")
    src = os.path.join(ROOT, "oracle", "oracle_loops.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield
