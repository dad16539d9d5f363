"""Build every native artefact in-tree (source-only history; the .so files travel to the GPU box).

  csrc/ggml + csrc/host      -> lib/libsdcpp-host.so      (g++   : graph front-end, model graph builders, sampler, C API)
  csrc/backend + csrc/kernels-> lib/libggml-mi355x.so     (hipcc : the MI355X ggml backend plug-in, gfx950 only)
  oracle/ggml_cpu_ref.cpp    -> oracle/_build/libggml-cpu-oracle.so (g++ -fopenmp : TEST ORACLE, never loaded by the product)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shlex
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
# SDCPP_BUILD_VARIANT=exp builds a second copy of the backend with -DMI355X_EXPERIMENTS (wrong-result timing ablations for
# scripts/*_ablation.py) into lib_exp/ — never loaded unless SDCPP_BACKEND_LIB points at it
VARIANT = os.environ.get("SDCPP_BUILD_VARIANT", "")
LIB = PKG / ("lib_" + VARIANT if VARIANT else "lib")
OBJ = PKG / ("build_" + VARIANT if VARIANT else "build")
ORACLE = ROOT / "oracle"
INCLUDE = ROOT / "include"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
ARCH = "gfx950"

HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
              f"-I{INCLUDE}", f"-I{CSRC / 'ggml'}", f"-I{CSRC}", "-pthread"]
HIP_FLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-Wall",
             "-Wno-unused-function", "-Wno-unused-result", "-Wno-pass-failed", f"-I{INCLUDE}", f"-I{CSRC}",
             f"-I{CSRC / 'backend'}", f"-I{CSRC / 'kernels'}", "-D__HIP_PLATFORM_AMD__"]
if VARIANT == "exp":
    HIP_FLAGS.append("-DMI355X_EXPERIMENTS")
if VARIANT.startswith("occ"):  # A/B: flash attention (d <= 64) compiled for occ4 / occ5 waves per SIMD (flash_attn.hip FA_OCC_SMALL)
    HIP_FLAGS.append("-DFA_OCC_SMALL=" + VARIANT[3:4])
ORACLE_FLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", "-fopenmp", "-mavx2", "-mfma", "-mf16c",
                "-Wall", f"-I{INCLUDE}"]


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write("FAILED: " + " ".join(shlex.quote(c) for c in cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build command failed: {cmd[0]} ... {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def _deps_hash(src: Path, flags: list[str], extra_dirs: list[Path]) -> str:
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    h.update(src.read_bytes())
    for d in extra_dirs:
        for p in sorted(d.rglob("*")):
            if p.suffix in (".h", ".hpp", ".cuh", ".hip.h"):
                h.update(p.read_bytes())
    return h.hexdigest()


def _compile(compiler: str, src: Path, flags: list[str], hdr_dirs: list[Path], tag: str = "") -> Path:
    OBJ.mkdir(exist_ok=True)
    obj = OBJ / (tag + src.parent.name + "_" + src.name + ".o")
    stamp = obj.with_suffix(".o.sha1")
    digest = _deps_hash(src, [compiler] + flags, hdr_dirs)
    if obj.exists() and stamp.exists() and stamp.read_text() == digest:
        return obj
    cmd = [compiler] + flags + ["-c", str(src), "-o", str(obj)]
    if compiler == HIPCC and src.suffix == ".cpp":
        cmd = [compiler] + flags + ["-x", "hip", "-c", str(src), "-o", str(obj)]
    _run(cmd)
    stamp.write_text(digest)
    return obj


def _link(compiler: str, objs: list[Path], out: Path, extra: list[str]) -> None:
    out.parent.mkdir(parents=True, exist_ok=True)
    newest = max(o.stat().st_mtime for o in objs)
    if out.exists() and out.stat().st_mtime >= newest:
        return
    _run([compiler, "-shared", "-o", str(out)] + [str(o) for o in objs] + extra)


def build_host(pool) -> Path:
    srcs = sorted((CSRC / "ggml").glob("*.cpp")) + sorted((CSRC / "host").glob("*.cpp"))
    hdrs = [CSRC / "ggml", CSRC / "host", INCLUDE]
    objs = list(pool.map(lambda s: _compile(CXX, s, HOST_FLAGS, hdrs), srcs))
    out = LIB / "libsdcpp-host.so"
    _link(CXX, objs, out, ["-ldl", "-pthread"])
    return out


def build_host_opshift(pool) -> Path:
    """TEST FIXTURE (tests/test_abi.py::test_op_enum_remap, tests/test_gpu_abi_remap.py): the same host built as a ggml "fork" whose op / unary-op
    enums have extra entries inserted mid-table (-DGGML_ABI_TEST_SHIFTED_ENUMS, include/ggml-abi.h).  The plug-in is NOT rebuilt: it has to find the
    shifted numbering through the host's ggml_op_name() at ggml_backend_init()."""
    srcs = sorted((CSRC / "ggml").glob("*.cpp")) + sorted((CSRC / "host").glob("*.cpp"))
    hdrs = [CSRC / "ggml", CSRC / "host", INCLUDE]
    flags = HOST_FLAGS + ["-DGGML_ABI_TEST_SHIFTED_ENUMS"]
    objs = list(pool.map(lambda s: _compile(CXX, s, flags, hdrs, tag="opshift_"), srcs))
    out = LIB / "libsdcpp-host-opshift.so"
    _link(CXX, objs, out, ["-ldl", "-pthread"])
    return out


def build_backend(pool) -> Path:
    srcs = sorted((CSRC / "backend").glob("*.cpp")) + sorted((CSRC / "kernels").glob("*.hip"))
    if not srcs:
        return LIB / "libggml-mi355x.so"
    hdrs = [CSRC / "backend", CSRC / "kernels", INCLUDE]
    objs = list(pool.map(lambda s: _compile(HIPCC, s, HIP_FLAGS, hdrs), srcs))
    out = LIB / "libggml-mi355x.so"
    _link(HIPCC, objs, out, [f"--offload-arch={ARCH}"])
    return out


def build_oracle(pool) -> Path:
    src = ORACLE / "ggml_cpu_ref.cpp"
    out = ORACLE / "_build" / "libggml-cpu-oracle.so"
    out.parent.mkdir(exist_ok=True)
    if out.exists() and out.stat().st_mtime >= max(src.stat().st_mtime, (INCLUDE / "ggml-abi.h").stat().st_mtime):
        return out
    _run([CXX] + ORACLE_FLAGS + ["-shared", str(src), "-o", str(out)])
    return out


def build_oracle_ref(pool):
    """oracle/_ref: the parts of the REFERENCE that compile from their own few sources where they lie (oracle/Makefile: PhiloxRNG, the CFG combine of
    src/runtime/guidance.cpp) — only where /root/reference exists (this container); the GPU box uses the prebuilt files that travel with the snapshot."""
    ref = Path(os.environ.get("SD_REFERENCE", "/root/reference"))
    if not (ref / "src" / "runtime" / "guidance.cpp").exists():
        return None
    _run(["make", "-C", str(ORACLE), "ref", f"REF={ref}"])
    return ORACLE / "_ref"


def build_all(verbose: bool = True) -> dict[str, Path]:
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        futs = {
            "host": pool.submit(build_host, pool),
            "backend": pool.submit(build_backend, pool),
            "oracle": pool.submit(build_oracle, pool),
            "host_opshift": pool.submit(build_host_opshift, pool),
            "oracle_ref": pool.submit(build_oracle_ref, pool),
        }
        out = {k: f.result() for k, f in futs.items() if f.result() is not None}
    if verbose:
        for k, v in out.items():
            print(f"[build] {k}: {v.relative_to(ROOT)}")
    return out


if __name__ == "__main__":
    build_all()
