"""ctypes front-end of oracle/rdx_oracle.c plus an independent numpy float64 restatement.

TEST INFRASTRUCTURE ONLY (see the header of rdx_oracle.c; parity of the top-k arithmetic with
chromadb==1.4.1 is UNPINNED — the wheel is absent, reference requirements.txt:32).

`-march=native`: the library is rebuilt on whichever host runs it (build() is called lazily), so a
.so built in the CPU container is never executed on a different micro-architecture by accident:
the build stamp records the host's CPU flags hash.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librdx_oracle.so")
_STAMP = os.path.join(_HERE, ".librdx_oracle.stamp")
_lib = None


def _host_tag() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()[:16]
    except OSError:
        pass
    return "unknown"


def build(force: bool = False) -> str:
    """Compile rdx_oracle.c with gcc (oracle/Makefile). Returns the .so path."""
    src = os.path.join(_HERE, "rdx_oracle.c")
    tag = _host_tag()
    fresh = (
        os.path.exists(_SO)
        and os.path.getmtime(_SO) >= os.path.getmtime(src)
        and os.path.exists(_STAMP)
        and open(_STAMP).read().strip() == tag
    )
    if force or not fresh:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "librdx_oracle.so"])
        with open(_STAMP, "w") as f:
            f.write(tag)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.rdxo_normalize_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.rdxo_normalize_rows.restype = None
        L.rdxo_score.argtypes = [f32p, f32p, ctypes.c_int]
        L.rdxo_score.restype = ctypes.c_float
        L.rdxo_scores.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, f32p]
        L.rdxo_scores.restype = None
        L.rdxo_cosine_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int64,
                                       ctypes.c_int, u32p, f32p, i64p, i32p]
        L.rdxo_cosine_topk.restype = None
        L.rdxo_merge_topk.argtypes = [f32p, i64p, i32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                      f32p, i64p, i32p]
        L.rdxo_merge_topk.restype = None
        L.rdxo_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def num_threads() -> int:
    return int(lib().rdxo_num_threads())


def normalize_rows(x) -> np.ndarray:
    """x / max(|x|, 1e-12) per row — reference src/utils/embedding_provider.py:139-145."""
    x = _f32(x)
    if x.ndim != 2 or x.shape[1] % 4:
        raise ValueError("normalize_rows wants [n][d] with d % 4 == 0")
    out = np.empty_like(x)
    lib().rdxo_normalize_rows(_p(x, ctypes.c_float), x.shape[0], x.shape[1], _p(out, ctypes.c_float))
    return out


def scores(corpus_hat, qhat) -> np.ndarray:
    corpus_hat = _f32(corpus_hat)
    qhat = _f32(qhat)
    out = np.empty(corpus_hat.shape[0], dtype=np.float32)
    lib().rdxo_scores(_p(corpus_hat, ctypes.c_float), corpus_hat.shape[0], corpus_hat.shape[1],
                      _p(qhat, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def pack_mask(allow: Optional[np.ndarray], n: int) -> Optional[np.ndarray]:
    """bool[n] -> uint32 words, bit (r & 31) of word r >> 5 (the layout include/rdx.h states)."""
    if allow is None:
        return None
    allow = np.asarray(allow, dtype=bool)
    assert allow.shape == (n,)
    padded = np.zeros(((n + 31) // 32) * 32, dtype=np.uint8)
    padded[:n] = allow
    return np.packbits(padded.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy()


def cosine_topk(corpus_hat, q_raw, k: int, allow: Optional[np.ndarray] = None
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """collection.query restated (reference src/rag/retriever.py:215-220). corpus_hat must already be
    normalised by normalize_rows. Returns (score f32[B,k], row i64[B,k], count i32[B])."""
    corpus_hat = _f32(corpus_hat)
    q_raw = _f32(q_raw)
    N, d = corpus_hat.shape
    B = q_raw.shape[0]
    if k < 0:
        raise ValueError("k must be >= 0")
    sc = np.empty((B, k), dtype=np.float32)
    ro = np.empty((B, k), dtype=np.int64)
    cn = np.empty((B,), dtype=np.int32)
    m = pack_mask(allow, N)
    lib().rdxo_cosine_topk(_p(corpus_hat, ctypes.c_float), N, d, _p(q_raw, ctypes.c_float), B, k,
                           _p(m, ctypes.c_uint32) if m is not None else None,
                           _p(sc, ctypes.c_float), _p(ro, ctypes.c_int64), _p(cn, ctypes.c_int32))
    return sc, ro, cn


def merge_topk(part_score, part_row, part_count, k: int):
    part_score = _f32(part_score)
    part_row = np.ascontiguousarray(part_row, dtype=np.int64)
    part_count = np.ascontiguousarray(part_count, dtype=np.int32)
    P, B, kk = part_score.shape
    assert kk == k
    sc = np.empty((B, k), dtype=np.float32)
    ro = np.empty((B, k), dtype=np.int64)
    cn = np.empty((B,), dtype=np.int32)
    lib().rdxo_merge_topk(_p(part_score, ctypes.c_float), _p(part_row, ctypes.c_int64),
                          _p(part_count, ctypes.c_int32), P, B, k,
                          _p(sc, ctypes.c_float), _p(ro, ctypes.c_int64), _p(cn, ctypes.c_int32))
    return sc, ro, cn


# ---- independent numpy restatement (float64 BLAS; summation order differs from the C oracle) -----

def normalize_numpy(x) -> np.ndarray:
    x64 = np.asarray(x, dtype=np.float64)
    den = np.maximum(np.sqrt((x64 * x64).sum(axis=1, keepdims=True)), 1e-12)
    return (x64 / den).astype(np.float32)


def topk_numpy(corpus_hat, q_raw, k: int, allow: Optional[np.ndarray] = None):
    """Same contract as cosine_topk, computed with float64 matmul + lexsort. Used to pin the C oracle:
    scores agree to 1 fp32 ulp, ids agree wherever the k-th/(k+1)-th gap exceeds that."""
    ch = np.asarray(corpus_hat, dtype=np.float64)
    qh = normalize_numpy(q_raw).astype(np.float64)
    s = (qh @ ch.T).astype(np.float32)
    N = ch.shape[0]
    B = qh.shape[0]
    sc = np.full((B, k), -np.inf, dtype=np.float32)
    ro = np.full((B, k), -1, dtype=np.int64)
    cn = np.zeros((B,), dtype=np.int32)
    rows = np.arange(N)
    for b in range(B):
        keep = rows if allow is None else rows[np.asarray(allow, dtype=bool)]
        order = keep[np.lexsort((keep, -s[b, keep].astype(np.float64)))][:k]
        sc[b, : len(order)] = s[b, order]
        ro[b, : len(order)] = order
        cn[b] = len(order)
    return sc, ro, cn


def topk_blas_f32(corpus_hat, qhat, k: int):
    """CPU throughput baseline (bench.py cpu_baseline): fp32 sgemm on all host cores + argpartition.
    Same semantics as the reference CPU path would have with an exact index; not bit-exact."""
    s = qhat @ corpus_hat.T
    part = np.argpartition(-s, kth=min(k, s.shape[1] - 1), axis=1)[:, :k]
    ps = np.take_along_axis(s, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    return np.take_along_axis(ps, order, axis=1), np.take_along_axis(part, order, axis=1)
