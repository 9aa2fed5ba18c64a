"""CPU: librdx.so loads and exports exactly the symbols include/rdx.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from rag_dpo_amd import _lib
    from rag_dpo_amd.build import build_lib
    build_lib()
    L = _lib.load(require_gpu=False)
    syms = declared_symbols()
    assert len(syms) >= 18 and "rdx_search" in syms and "rdx_merge_topk_packed" in syms
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/rdx.h but not exported by librdx.so"
    assert sorted(_lib.SYMBOLS) == syms, "ctypes binding table and header disagree"
    assert L.rdx_version() == _lib.ABI_VERSION == 3
    assert ctypes.sizeof(_lib.SearchStats) == 8 * 7 + 4 * 2 + 4 * 7 + 4 + 8 * 3 + 4 * 4   # layout of rdx_search_stats (with padding)


def test_error_convention_without_gpu():
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load(require_gpu=False)
    n = ctypes.c_int(-1)
    rc = L.rdx_device_count(ctypes.byref(n))
    if not torch.cuda.is_available():
        assert rc != 0 and _lib.last_error() != ""       # failure -> code + thread-local message, never a crash
        h = ctypes.c_void_p()
        assert L.rdx_index_create(0, 1024, ctypes.byref(h)) != 0
    assert L.rdx_index_create(0, 1023, ctypes.byref(ctypes.c_void_p())) == _lib.RDX_ERR_INVALID   # dim % 4
    assert "multiple of 4" in _lib.last_error()


def test_scan_kernels_do_not_spill():
    """the scan kernels keep inline-asm loads in flight across many instructions: a register spill in them can save and
    restore a value that has not landed yet (silently wrong ids). The build refuses such a library; this pins the
    figures it recorded (rag_dpo_amd/build.py)."""
    import json
    from rag_dpo_amd import build
    build.build_lib()
    res = json.load(open(build.RESOURCES))
    assert res["_build"]["source_sha256"] == build.source_hash()
    scans = {k: v for k, v in res.items() if "k_scan" in k}
    assert len(scans) >= 16, sorted(scans)
    for k, v in scans.items():
        assert v["spill_vgprs"] == 0 and v["scratch_bytes"] == 0 and v["vgprs"] <= 256, (k, v)
