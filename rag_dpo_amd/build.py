"""Builds librdx.so (HIP, gfx950) in-tree. hipcc cross-compiles without a GPU.

The scan kernel issues its corpus loads as inline asm with hand-counted waits (scan_kernel.hpp): a register such a load
is still in flight to must never be spilled or copied by the compiler, which it would do silently under register
pressure (seen once: 2 % wrong ids in one variant, caught by the parity tests). The build therefore reads the
compiler's own resource remarks and REFUSES a library whose scan kernels spill; the figures are kept next to the
library (librdx.resources.json) and checked again by tests/test_abi.py."""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdx.so")
RESOURCES = os.path.join(HERE, "librdx.resources.json")
SOURCES = ["rdx_api.hip"]
HEADERS = ["rdx_common.hpp", "k_rows.hpp", "scan_kernel.hpp", "refine_kernel.hpp", "../../include/rdx.h"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librdx needs the ROCm toolchain (/opt/rocm/bin/hipcc)")


def source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_fresh() -> bool:
    """the library was built from exactly the sources in the tree (content hash recorded next to it; mtimes alone
    would accept a variant library copied over the product file)"""
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return False
    try:
        with open(RESOURCES) as f:
            rec = json.load(f).get("_build", {})
    except (OSError, ValueError):
        return False
    return rec.get("source_sha256") == source_hash() and rec.get("lib_size") == os.path.getsize(LIB)


def parse_resources(remarks: str) -> dict:
    """{kernel symbol: {"vgprs", "agprs", "spill_vgprs", "spill_sgprs", "scratch_bytes", "lds_bytes"}} from
    -Rpass-analysis=kernel-resource-usage output"""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "VGPRs Spill": "spill_vgprs", "SGPRs Spill": "spill_sgprs",
            "ScratchSize [bytes/lane]": "scratch_bytes", "LDS Size [bytes/block]": "lds_bytes"}
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """out: build a VARIANT (extra_flags) to this path; the product library and its resource record are left alone"""
    if out is None and not force and is_fresh() and not extra_flags:
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra_flags]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    tmp = (out or LIB) + ".tmp"
    cmd += ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    noise = ("-Rpass-analysis=kernel-resource-usage", "remark:")
    diag = "\n".join(l for l in r.stderr.splitlines() if not any(n in l for n in noise))
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{diag[-4000:]}")
    res = parse_resources(r.stderr)
    bad = {k: v for k, v in res.items() if "k_scan" in k and (v.get("spill_vgprs", 0) or v.get("scratch_bytes", 0))}
    if bad:
        os.remove(tmp)
        raise RuntimeError("scan kernels spill registers — refused (inline-asm loads may be in flight to a spilled register):\n" +
                           "\n".join(f"  {k}: {v}" for k, v in bad.items()))
    if out is not None:
        os.replace(tmp, out)
        return out
    os.replace(tmp, LIB)
    res["_build"] = {"source_sha256": source_hash(), "lib_size": os.path.getsize(LIB), "flags": list(extra_flags)}
    with open(RESOURCES, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    if verbose and diag.strip():
        print(diag)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
