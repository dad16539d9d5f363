import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# the CPU oracle is OpenMP code: keep its team small and passive (GPU boxes expose hundreds of hardware threads to
# containers that can schedule only a few; a spinning oversubscribed team makes whole-graph oracle runs crawl)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 1, 16))))
if (os.cpu_count() or 1) > 32:  # big shared host: do not spin-wait an oversubscribed team
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
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
    if os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE"):  # harness self-check on a CPU-only box: both sides run the oracle
        sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
        return "CPU-oracle"
    sd.load_mi355x_backend()
    # debugging aid: SDCPP_BACKEND_OPTS="fuse_gate=0,gemm16_tile=1" applies planner options for the whole session
    for kv in filter(None, os.environ.get("SDCPP_BACKEND_OPTS", "").split(",")):
        k, v = kv.split("=")
        sd.backend_set_option(k.strip(), int(v))
    return "MI355X0"
