"""Import shim: the package directory is `stable-diffusion.cpp_amd/` (not a valid Python identifier), so
`import sdcpp_amd` loads it under this name (submodules resolve as `sdcpp_amd.build`, ...)."""
import importlib.util
import pathlib
import sys

_pkg = pathlib.Path(__file__).resolve().parent / "stable-diffusion.cpp_amd"
_spec = importlib.util.spec_from_file_location("sdcpp_amd", _pkg / "__init__.py", submodule_search_locations=[str(_pkg)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sdcpp_amd"] = _mod
_spec.loader.exec_module(_mod)
