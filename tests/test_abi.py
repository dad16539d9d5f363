"""CPU suite: the C-ABI libraries load and export every symbol include/*.h declares (no compute without a GPU)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def exported(so: Path) -> set:
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def declared(header: Path, macro: str) -> set:
    txt = header.read_text()
    return set(re.findall(macro + r"\s+[\w\s\*]+?\b(\w+)\s*\(", txt))


def test_backend_plugin_exports(sd):
    so = sd.BACKEND_LIB
    assert so.exists(), "libggml-mi355x.so was not built"
    want = declared(ROOT / "include" / "ggml-mi355x.h", "GGML_MI355X_API")
    assert {"ggml_backend_init", "ggml_backend_score", "ggml_backend_mi355x_reg"} <= want
    missing = want - exported(so)
    assert not missing, missing


def test_backend_plugin_loads_without_gpu_and_reports_no_device(sd):
    lib = C.CDLL(str(sd.BACKEND_LIB))
    lib.ggml_backend_score.restype = C.c_int
    lib.ggml_backend_mi355x_get_device_count.restype = C.c_int
    n = lib.ggml_backend_mi355x_get_device_count()
    assert lib.ggml_backend_score() == (100 if n > 0 else 0)
    lib.ggml_backend_init.restype = C.c_void_p
    reg = lib.ggml_backend_init()
    assert reg  # registry exists even with zero devices; api_version is its first field
    assert C.cast(reg, C.POINTER(C.c_int)).contents.value == 2


def test_host_library_exports(sd):
    want = declared(ROOT / "include" / "sd-mi355x.h", "SD_API")
    assert {"sdm_new_ctx", "sdm_generate_image", "sdm_free_images", "sd_unet_forward", "sd_vae_decode"} <= want
    missing = want - exported(sd.HOST_LIB)
    assert not missing, missing


def test_engine_refuses_to_run_without_the_hip_backend(sd):
    """The product path must fail loudly (no CPU fallback) when no MI355X device is registered."""
    import pytest

    if any(d.startswith("MI355X") for d in sd.devices()):
        pytest.skip("a GPU is present")
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except Exception:
        pass
    with pytest.raises(sd.EngineError):
        sd.Engine(model=sd.SD15_TINY)  # default backend = MI355X0
