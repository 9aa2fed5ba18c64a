"""Builds librdx.so (HIP, gfx950) in-tree. hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdx.so")
SOURCES = ["rdx_api.hip"]
HEADERS = ["rdx_common.hpp", "k_rows.hpp", "scan_kernel.hpp", "refine_kernel.hpp", "../../include/rdx.h"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librdx needs the ROCm toolchain (/opt/rocm/bin/hipcc)")


def is_fresh() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return all(os.path.getmtime(d) <= t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and is_fresh():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
