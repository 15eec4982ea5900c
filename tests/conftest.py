import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("PARESIS_ALLOW_SYNTHETIC_MATERIALS", "1")   # the shipped XML materials resolve to the synthetic stand-ins here


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle's C loops are test infrastructure; build them on demand (gcc is in the image)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle_loops.so")
    src = os.path.join(ROOT, "oracle", "oracle_loops.c")
    so2 = os.path.join(ROOT, "oracle", "libcpu_baseline.so")
    src2 = os.path.join(ROOT, "oracle", "cpu_baseline.cpp")
    if (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) or not os.path.exists(so2)
            or os.path.getmtime(so2) < os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield


@pytest.fixture(scope="session", autouse=True)
def _library_built():
    """libparesis_hip.so is git-ignored: on a fresh checkout build it once (hipcc cross-compiles gfx950 without a GPU).
    Only when it is missing -- a present library is never rebuilt behind the tests' back."""
    import subprocess
    so = os.path.join(ROOT, "paresis_amd", "libparesis_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "paresis_amd", "csrc"), "-j8"])
    yield
