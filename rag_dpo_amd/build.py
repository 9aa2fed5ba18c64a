"""Builds librdx.so (HIP, gfx950) in-tree. hipcc cross-compiles without a GPU.

The scan kernel issues its corpus loads as inline asm with hand-counted waits (scan_kernel.hpp): a register such a load
is still in flight to must never be spilled or copied by the compiler, which it would do silently under register
pressure (seen once: 2 % wrong ids in one variant, caught by the parity tests). The build therefore reads the
compiler's own resource remarks and REFUSES a library whose scan kernels spill; the figures are kept next to the
library (librdx.resources.json) and checked again by tests/test_abi.py.

"No spills" is one way of breaking that property, not the property. Since round 3 the build also reads the ISA it is about to
ship (-save-temps: the very listing that is assembled into the library) and refuses a library in which, on any path through a
k_scan kernel, (A) anything but the source's own MFMA reads and asm loads touches the destination registers of an asm load
that may still be in flight, (B) a fragment is used before its load can have retired on any path, or (C) an inline-asm memory
instruction reads an SGPR fewer than 5 wait states after a VALU instruction wrote it (the hazard the compiler cannot see
through inline asm; it is what made the first RDX_CHECK_BOUNDS build fault) — rag_dpo_amd/isa_check.py."""
from __future__ import annotations

import glob
import json
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdx.so")
RESOURCES = os.path.join(HERE, "librdx.resources.json")
SOURCES = ["rdx_api.hip"]
HEADERS = ["rdx_common.hpp", "k_rows.hpp", "scan_kernel.hpp", "scan_w4.hpp", "refine_kernel.hpp", "enc_kernels.hpp", "enc_small.hpp", "../../include/rdx.h"]
CHECKERS = ["isa_check.py"]   # part of the recorded hash: a library is only "fresh" if it passed THIS version of the ISA check


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
    for f in CHECKERS:
        with open(os.path.join(HERE, f), "rb") as fh:
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
        if "remark" not in line:
            continue
        # two layouts: "<file>:<line>:<col>: remark: Function Name: X" and, with -save-temps, "remark: <file>:<line>:<col>: Function Name: X"
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r":\s+([A-Za-z][A-Za-z \[\]/]*): (\d+)(?: \[-Rpass|$)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def check_isa(work: str) -> dict:
    """run rag_dpo_amd/isa_check.py over the device listing -save-temps left in `work`; raises on any hazard"""
    from . import isa_check
    lst = glob.glob(os.path.join(work, "*gfx950*.s"))
    if len(lst) != 1:
        raise RuntimeError(f"expected one gfx950 listing from -save-temps in {work}, found {lst}")
    found = isa_check.check_listing(open(lst[0]).read())
    if len(found) < 16:
        raise RuntimeError(f"ISA check: only {len(found)} k_scan kernels in the listing — the parser no longer matches the compiler's output")
    bad = {k: hz for k, (hz, st) in found.items() if hz}
    if bad:
        lines = []
        for k, hz in bad.items():
            lines.append(f"  {k}: {len(hz)} hazard(s)")
            lines += [f"    [{h['kind']}] line {h['line']}: {h['text']}  <->  line {h['load_line']}: {h['load_text']}" for h in hz[:6]]
        raise RuntimeError("scan kernels fail the ISA check — refused (rag_dpo_amd/isa_check.py):\n" + "\n".join(lines))
    return {"kernels": len(found), "asm_loads": sum(st["asm_loads"] for _, st in found.values()),
            "asm_vmem": sum(st["asm_vmem"] for _, st in found.values()), "hazards": 0,
            "checks": ["A: no foreign instruction touches an in-flight fragment", "B: no fragment used before its load can have retired",
                       "C: 5 wait states between a VALU SGPR write and an asm VMEM read of it"]}


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """out: build a VARIANT (extra_flags) to this path; the product library and its resource record are left alone"""
    if out is None and not force and is_fresh() and not extra_flags:
        return LIB
    # compiled in a scratch directory with -save-temps=obj: the device listing (…gfx950.s) lands beside the output and is the
    # very text that gets assembled into the library
    work = tempfile.mkdtemp(prefix="librdx_build_", dir=os.environ.get("TMPDIR") or None)
    try:
        tmp = os.path.join(work, "librdx.so")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-save-temps=obj",
               "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra_flags]
        cmd += [os.path.join(CSRC, s) for s in SOURCES]
        cmd += ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=work)
        noise = ("-Rpass-analysis=kernel-resource-usage", "remark:")
        diag = "\n".join(l for l in r.stderr.splitlines() if not any(n in l for n in noise))
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}):\n{diag[-4000:]}")
        res = parse_resources(r.stderr)
        if sum(1 for k, v in res.items() if "k_scan" in k and "spill_vgprs" in v and "scratch_bytes" in v) < 16:
            raise RuntimeError("the compiler's resource remarks could not be read for the scan kernels — refused (the spill check would be blind)")
        bad = {k: v for k, v in res.items() if "k_scan" in k and (v.get("spill_vgprs", 0) or v.get("scratch_bytes", 0))}
        if bad:
            raise RuntimeError("scan kernels spill registers — refused (inline-asm loads may be in flight to a spilled register):\n" +
                               "\n".join(f"  {k}: {v}" for k, v in bad.items()))
        # kernels that are given an occupancy target (amdgpu_waves_per_eu: the latency-bound attention kernel E12) pay for a miss
        # silently, in scratch traffic: refused as well (a variant build with extra flags may spill: it is nobody's product)
        slow = {k: v for k, v in res.items() if "k_enc_attention_mfma" in k and (v.get("spill_vgprs", 0) or v.get("scratch_bytes", 0))}
        if slow and not extra_flags:
            raise RuntimeError("k_enc_attention_mfma no longer fits its occupancy target without spilling — refused:\n" +
                               "\n".join(f"  {k}: {v}" for k, v in slow.items()))
        isa = check_isa(work)
        final_tmp = (out or LIB) + ".tmp"
        shutil.move(tmp, final_tmp)
        tmp = final_tmp
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if out is not None:
        os.replace(tmp, out)
        return out
    os.replace(tmp, LIB)
    res["_build"] = {"source_sha256": source_hash(), "lib_size": os.path.getsize(LIB), "flags": list(extra_flags)}
    res["_isa_check"] = isa
    with open(RESOURCES, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    if verbose and diag.strip():
        print(diag)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
