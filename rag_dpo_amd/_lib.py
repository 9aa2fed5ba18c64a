"""ctypes binding of librdx.so (include/rdx.h). There is no CPU fallback: a missing or unloadable
library, or a machine without a gfx950 GPU, raises RdxUnavailable as soon as the hot path is used."""
from __future__ import annotations

import ctypes
import os

# torch bundles its own libamdhip64 (same soname as ROCm's): import it first so that librdx and
# torch.distributed/RCCL share ONE HIP runtime in the process.
import torch  # noqa: F401  (plumbing only: device tensors, streams, torch.distributed)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDX_LIB_PATH: developer override (tools/ab_bench.sh points a run at a variant build WITHOUT touching the product file)
LIB_PATH = os.environ.get("RDX_LIB_PATH") or os.path.join(_HERE, "librdx.so")

RDX_OK, RDX_ERR_INVALID, RDX_ERR_HIP, RDX_ERR_NOMEM, RDX_ERR_STATE = 0, 1, 2, 3, 4
RDX_HOST, RDX_DEVICE = 0, 1
ABI_VERSION = 3          # include/rdx.h RDX_ABI_VERSION
PACKED_FLAGS = 4         # include/rdx.h RDX_PACKED_FLAGS: int32 words behind the counts of a packed partial


class RdxUnavailable(RuntimeError):
    """The HIP library is not built/loadable or no gfx950 device is visible."""


class RdxError(RuntimeError):
    pass


class SearchStats(ctypes.Structure):
    _fields_ = [
        ("nq", ctypes.c_int64), ("k", ctypes.c_int64), ("rows", ctypes.c_int64),
        ("sample_rows", ctypes.c_int64), ("emitted", ctypes.c_int64), ("rescored", ctypes.c_int64),
        ("exact_queries", ctypes.c_int64), ("path", ctypes.c_int32), ("profiled", ctypes.c_int32),
        ("ms_normalize", ctypes.c_float), ("ms_scan_sample", ctypes.c_float), ("ms_tau", ctypes.c_float),
        ("ms_scan_main", ctypes.c_float), ("ms_refine", ctypes.c_float), ("ms_exact", ctypes.c_float),
        ("ms_total", ctypes.c_float),
        ("scan_main_launch_rows", ctypes.c_int64), ("scan_main_launch_queries", ctypes.c_int64),
        ("retried_queries", ctypes.c_int64),
        ("xcd_finish_spread_ms", ctypes.c_float), ("xcd_share_min", ctypes.c_float), ("xcd_share_max", ctypes.c_float),
        ("tau_rank", ctypes.c_float),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None

# every symbol include/rdx.h declares: (restype, argtypes)
_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
SYMBOLS = {
    "rdx_version": (_i, []),
    "rdx_last_error": (ctypes.c_char_p, []),
    "rdx_device_count": (_i, [ctypes.POINTER(_i)]),
    "rdx_set_wait_policy": (_i, [_i, _i]),
    "rdx_index_create": (_i, [_i, _i, ctypes.POINTER(_vp)]),
    "rdx_index_destroy": (_i, [_vp]),
    "rdx_index_dim": (_i, [_vp, ctypes.POINTER(_i)]),
    "rdx_index_count": (_i, [_vp, ctypes.POINTER(_i64)]),
    "rdx_index_reserve": (_i, [_vp, _i64]),
    "rdx_index_add": (_i, [_vp, _vp, _i64, _i]),
    "rdx_index_add_bf16": (_i, [_vp, _vp, _i64, _i]),
    "rdx_index_add_stored": (_i, [_vp, _vp, _i64, _i]),
    "rdx_index_update": (_i, [_vp, _vp, _vp, _i64, _i]),
    "rdx_index_get": (_i, [_vp, _vp, _i64, _vp, _i]),
    "rdx_index_compact": (_i, [_vp, _vp, _i64]),
    "rdx_index_set_row_ids": (_i, [_vp, _i64, _vp, _i64, _i]),
    "rdx_index_set_option": (_i, [_vp, ctypes.c_char_p, _i64]),
    "rdx_index_xcd_shares": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "rdx_l2_normalize": (_i, [_i, _vp, _i64, _i, _vp, _i, _vp]),
    "rdx_enc_attention_f16": (_i, [_i, _vp, _vp, _vp, _i64, _i, _i, ctypes.c_float, _i, _vp, _vp]),
    "rdx_enc_attention_mfma_f16": (_i, [_i, _vp, _vp, _i, _i, _i, ctypes.c_float, _vp, _vp]),
    "rdx_enc_gelu_f16": (_i, [_i, _vp, ctypes.c_int64, _vp]),
    "rdx_enc_linear_small_f16": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rdx_enc_add_layernorm_f16": (_i, [_i, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i, _vp, _vp]),
    "rdx_enc_embed_f16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "rdx_enc_stage_f16": (_i, [_i, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rdx_enc_attention_small_f16": (_i, [_i, _vp, _vp, _i, _i, _i, ctypes.c_float, _vp, _vp]),
    "rdx_enc_layernorm_rows_f16": (_i, [_i, _vp, _vp, _vp, ctypes.c_float, _i, _i, _vp, _vp]),
    "rdx_search": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "rdx_mask_create": (_i, [_vp, _vp, _i, ctypes.POINTER(_vp)]),
    "rdx_mask_destroy": (_i, [_vp]),
    "rdx_search_masked": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "rdx_search_async": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdx_search_wait": (_i, [_vp, ctypes.POINTER(_i)]),
    "rdx_merge_topk": (_i, [_i, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _i, _vp]),
    "rdx_signal_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "rdx_signal_destroy": (_i, [_vp]),
    "rdx_signal_wait": (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_int32)]),
    "rdx_merge_topk_packed": (_i, [_i, _vp, _i64, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdx_search_last_stats": (_i, [_vp, ctypes.POINTER(SearchStats)]),
}


def load(require_gpu: bool = True):
    """Load librdx.so. Raises RdxUnavailable (never falls back) when it cannot serve the hot path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RdxUnavailable(
                f"{LIB_PATH} is missing: build it with `python -m rag_dpo_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the retrieval hot path.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RdxUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.rdx_version() != ABI_VERSION:
            raise RdxUnavailable(f"librdx ABI version {L.rdx_version()} != {ABI_VERSION}: rebuild the library")
        _lib = L
    if require_gpu:
        n = ctypes.c_int(0)
        rc = _lib.rdx_device_count(ctypes.byref(n))
        if rc != RDX_OK or n.value < 1:
            raise RdxUnavailable("no HIP device visible: the retrieval hot path needs an MI355X (gfx950) GPU; "
                                 "there is no CPU fallback. " + last_error())
    return _lib


def last_error() -> str:
    if _lib is None:
        return ""
    return (_lib.rdx_last_error() or b"").decode("utf-8", "replace")


def check(rc: int):
    if rc == RDX_OK:
        return
    msg = last_error()
    if rc == RDX_ERR_INVALID:
        raise ValueError(msg)
    if rc == RDX_ERR_NOMEM:
        raise MemoryError(msg)
    raise RdxError(f"librdx error {rc}: {msg}")
