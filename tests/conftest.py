import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sd():
    """The package (ctypes binding), with native libs built."""
    import sdcpp_amd
    from sdcpp_amd import build

    if not sdcpp_amd.HOST_LIB.exists() or os.environ.get("SDCPP_REBUILD"):
        build.build_all(verbose=False)
    sdcpp_amd.lib()
    return sdcpp_amd


@pytest.fixture(scope="session")
def oracle(sd):
    """Registers the CPU oracle backend plug-in (TEST infrastructure) and returns its device name."""
    p = ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so"
    if not p.exists():
        from sdcpp_amd import build
        build.build_all(verbose=False)
    sd.load_backend(p)
    return "CPU-oracle"


@pytest.fixture(scope="session")
def gpu(sd):
    """Loads the product backend; fails loudly if the HIP extension or the GPU is missing."""
    sd.load_mi355x_backend()
    return "MI355X0"
