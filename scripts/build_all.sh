#!/bin/bash
# rebuild every native artefact in-tree (stable-diffusion.cpp_amd/build.py)
cd "$(dirname "$0")/.." && python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','stable-diffusion.cpp_amd/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build_all()" 2>&1 | tail -${1:-3}
