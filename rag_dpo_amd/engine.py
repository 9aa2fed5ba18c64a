"""HipIndex — one corpus shard in one MI355X's HBM, driven through the C-ABI (include/rdx.h).

numpy arrays cross as host pointers (RDX_HOST, synchronous); torch CUDA tensors cross as device
pointers (RDX_DEVICE) on the current torch stream. No arithmetic happens in this file.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _lib as L


def _np_ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


class ResidentMask:
    """a `where` bitmap resident in HBM (rdx_mask): made once per distinct filter, passed with every search"""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.rdx_mask_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipIndex:
    def __init__(self, dim: int, device: int = 0):
        self._lib = L.load(require_gpu=True)
        self._h = ctypes.c_void_p()
        L.check(self._lib.rdx_index_create(int(device), int(dim), ctypes.byref(self._h)))
        self.dim = int(dim)
        self.device = int(device)

    # ---- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rdx_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = ctypes.c_int64(0)
        L.check(self._lib.rdx_index_count(self._h, ctypes.byref(n)))
        return int(n.value)

    def reserve(self, rows: int):
        L.check(self._lib.rdx_index_reserve(self._h, int(rows)))

    def set_option(self, name: str, value: int):
        L.check(self._lib.rdx_index_set_option(self._h, name.encode(), int(value)))

    def xcd_shares(self, new=None):
        """the main scan's per-XCD tile shares (include/rdx.h rdx_index_xcd_shares): returns the current 8 values; `new` replaces them"""
        out = (ctypes.c_double * 8)()
        inp = (ctypes.c_double * 8)(*[float(x) for x in new]) if new is not None else None
        L.check(self._lib.rdx_index_xcd_shares(self._h, out, inp))
        return list(out)

    # ---- helpers ----------------------------------------------------------------------------
    def _rows_arg(self, x, dtype=np.float32):
        """-> (pointer, n, space, keepalive)"""
        import torch
        if isinstance(x, torch.Tensor):
            want = torch.float32 if dtype == np.float32 else torch.bfloat16
            if x.dtype != want or x.dim() != 2 or x.shape[1] != self.dim:
                raise ValueError(f"expected a [n][{self.dim}] {want} tensor, got {tuple(x.shape)} {x.dtype}")
            if x.is_cuda:
                if x.device.index != self.device:
                    raise ValueError("tensor lives on another device than the index")
                x = x.contiguous()
                torch.cuda.current_stream(x.device).synchronize()  # ingest runs on the index's own stream
                return ctypes.c_void_p(x.data_ptr()), x.shape[0], L.RDX_DEVICE, x
            x = x.numpy() if dtype == np.float32 else x.view(torch.uint16).numpy()
        a = np.ascontiguousarray(x, dtype=dtype if dtype == np.float32 else np.uint16)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError(f"expected [n][{self.dim}] embeddings, got shape {a.shape}")
        return _np_ptr(a), a.shape[0], L.RDX_HOST, a

    # ---- ingest -----------------------------------------------------------------------------
    def add(self, rows):
        p, n, space, keep = self._rows_arg(rows)
        L.check(self._lib.rdx_index_add(self._h, p, n, space))

    def add_stored(self, rows):
        """rows previously returned by get() (already normalised): stored verbatim (snapshot reload)"""
        p, n, space, keep = self._rows_arg(rows)
        L.check(self._lib.rdx_index_add_stored(self._h, p, n, space))

    def add_bf16(self, rows):
        p, n, space, keep = self._rows_arg(rows, dtype=np.uint16)
        L.check(self._lib.rdx_index_add_bf16(self._h, p, n, space))

    def update(self, row_ids, rows):
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        a = np.ascontiguousarray(rows, dtype=np.float32)
        if a.ndim != 2 or a.shape != (ids.shape[0], self.dim):
            raise ValueError("update: rows must be [len(row_ids)][dim]")
        L.check(self._lib.rdx_index_update(self._h, _np_ptr(ids), _np_ptr(a), ids.shape[0], L.RDX_HOST))

    def get(self, row_ids) -> np.ndarray:
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        out = np.empty((ids.shape[0], self.dim), dtype=np.float32)
        L.check(self._lib.rdx_index_get(self._h, _np_ptr(ids), ids.shape[0], _np_ptr(out), L.RDX_HOST))
        return out

    def set_row_ids(self, first_row: int, ids):
        """returned row id of local row first_row + i = ids[i] (strictly increasing over the shard)"""
        a = np.ascontiguousarray(ids, dtype=np.int64)
        L.check(self._lib.rdx_index_set_row_ids(self._h, int(first_row), _np_ptr(a), a.shape[0], L.RDX_HOST))

    def compact(self, keep_rows):
        keep = np.ascontiguousarray(keep_rows, dtype=np.int64)
        L.check(self._lib.rdx_index_compact(self._h, _np_ptr(keep), keep.shape[0]))

    # ---- search -----------------------------------------------------------------------------
    def make_mask(self, allow_bits: np.ndarray) -> ResidentMask:
        allow_bits = np.ascontiguousarray(allow_bits, dtype=np.uint32)
        if allow_bits.shape[0] != (len(self) + 31) // 32:
            raise ValueError("allow_bits must hold ceil(count/32) words")
        h = ctypes.c_void_p()
        L.check(self._lib.rdx_mask_create(self._h, _np_ptr(allow_bits), L.RDX_HOST, ctypes.byref(h)))
        return ResidentMask(self._lib, h)

    def search(self, queries, k: int, allow_bits: Optional[np.ndarray] = None, mask: Optional[ResidentMask] = None
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Host in / host out. Returns (score f32[nq,k], row i64[nq,k], count i32[nq]).
        allow_bits: host bitmap uploaded for this call; mask: a resident one (make_mask) — not both."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq][{self.dim}] query embeddings, got shape {q.shape}")
        nq = q.shape[0]
        sc = np.empty((nq, k), dtype=np.float32)
        ro = np.empty((nq, k), dtype=np.int64)
        cn = np.zeros((nq,), dtype=np.int32)
        if mask is not None:
            if allow_bits is not None:
                raise ValueError("pass allow_bits or mask, not both")
            L.check(self._lib.rdx_search_masked(self._h, _np_ptr(q), nq, int(k), mask._h, _np_ptr(sc), _np_ptr(ro), _np_ptr(cn),
                                                L.RDX_HOST, None))
            return sc, ro, cn
        mp = None
        if allow_bits is not None:
            allow_bits = np.ascontiguousarray(allow_bits, dtype=np.uint32)
            if allow_bits.shape[0] != (len(self) + 31) // 32:
                raise ValueError("allow_bits must hold ceil(count/32) words")
            mp = _np_ptr(allow_bits)
        L.check(self._lib.rdx_search(self._h, _np_ptr(q), nq, int(k), mp, _np_ptr(sc), _np_ptr(ro), _np_ptr(cn),
                                     L.RDX_HOST, None))
        return sc, ro, cn

    @staticmethod
    def _raw_stream(device) -> int:
        """handle of torch's current stream on `device` (the private one-call form when this torch has it: 0.3 instead of 4 us)"""
        import torch
        try:
            return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
        except AttributeError:   # pragma: no cover
            return torch.cuda.current_stream(device).cuda_stream

    def search_device(self, queries, k: int, out_score, out_row, out_count, allow_bits=None, mask: Optional[ResidentMask] = None):
        """torch CUDA tensors in/out, enqueued on the current torch stream (no host copies).
        allow_bits: a device bitmap for this call; mask: a resident one (make_mask) — not both."""
        import torch
        nq = queries.shape[0]
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        assert out_score.shape == (nq, k) and out_score.dtype == torch.float32 and out_score.is_contiguous()
        assert out_row.shape == (nq, k) and out_row.dtype == torch.int64 and out_row.is_contiguous()
        assert out_count.shape == (nq,) and out_count.dtype == torch.int32
        stream = self._raw_stream(queries.device)
        if mask is not None:
            if allow_bits is not None:
                raise ValueError("pass allow_bits or mask, not both")
            L.check(self._lib.rdx_search_masked(self._h, ctypes.c_void_p(queries.data_ptr()), nq, int(k), mask._h,
                                                ctypes.c_void_p(out_score.data_ptr()), ctypes.c_void_p(out_row.data_ptr()),
                                                ctypes.c_void_p(out_count.data_ptr()), L.RDX_DEVICE, ctypes.c_void_p(stream)))
            return
        mp = ctypes.c_void_p(allow_bits.data_ptr()) if allow_bits is not None else None
        L.check(self._lib.rdx_search(self._h, ctypes.c_void_p(queries.data_ptr()), nq, int(k), mp,
                                     ctypes.c_void_p(out_score.data_ptr()), ctypes.c_void_p(out_row.data_ptr()),
                                     ctypes.c_void_p(out_count.data_ptr()), L.RDX_DEVICE, ctypes.c_void_p(stream)))

    def search_device_async(self, queries, k: int, out_score, out_row, out_count, out_flags=None, mask: Optional[ResidentMask] = None):
        """enqueue the search on the current torch stream and return at once; search_wait() completes it (include/rdx.h).
        `queries` may be reused by stream-ordered work enqueued afterwards; the outputs (and out_flags: int32[4] on the device,
        [0] = 1 while the results are incomplete) must stay valid until search_wait() has returned."""
        import torch
        nq = queries.shape[0]
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        assert out_score.shape == (nq, k) and out_score.dtype == torch.float32 and out_score.is_contiguous()
        assert out_row.shape == (nq, k) and out_row.dtype == torch.int64 and out_row.is_contiguous()
        assert out_count.shape == (nq,) and out_count.dtype == torch.int32
        assert out_flags is None or (out_flags.numel() >= L.PACKED_FLAGS and out_flags.dtype == torch.int32)
        stream = self._raw_stream(queries.device)
        L.check(self._lib.rdx_search_async(self._h, ctypes.c_void_p(queries.data_ptr()), nq, int(k), mask._h if mask is not None else None,
                                           ctypes.c_void_p(out_score.data_ptr()), ctypes.c_void_p(out_row.data_ptr()),
                                           ctypes.c_void_p(out_count.data_ptr()),
                                           ctypes.c_void_p(out_flags.data_ptr()) if out_flags is not None else None,
                                           ctypes.c_void_p(stream)))

    def search_wait(self) -> bool:
        """-> True when fallback passes rewrote results after the asynchronous search's kernels (consumers must be re-run)"""
        redone = ctypes.c_int(0)
        L.check(self._lib.rdx_search_wait(self._h, ctypes.byref(redone)))
        return bool(redone.value)

    def last_stats_struct(self):
        """the raw rdx_search_stats of the last search (a ctypes struct: field access without building a dict)"""
        s = L.SearchStats()
        L.check(self._lib.rdx_search_last_stats(self._h, ctypes.byref(s)))
        return s

    def last_stats(self) -> dict:
        s = L.SearchStats()
        L.check(self._lib.rdx_search_last_stats(self._h, ctypes.byref(s)))
        return s.as_dict()


def l2_normalize(x: np.ndarray, device: int = 0) -> np.ndarray:
    lib = L.load(require_gpu=True)
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("l2_normalize wants [n][dim]")
    out = np.empty_like(a)
    L.check(lib.rdx_l2_normalize(device, _np_ptr(a), a.shape[0], a.shape[1], _np_ptr(out), L.RDX_HOST, None))
    return out


def merge_topk_device(part_score, part_row, part_count, k: int, out_score, out_row, out_count):
    """rdx_merge_topk on torch CUDA tensors ([P][nq][k], [P][nq][k], [P][nq] on ONE device), enqueued on that device's current
    torch stream; any P * k (the library folds the parts pairwise beyond what one merge launch ranks)"""
    lib = L.load(require_gpu=True)
    P, nq = part_count.shape
    dev = part_score.device
    assert part_score.is_contiguous() and part_row.is_contiguous() and part_count.is_contiguous()
    stream = HipIndex._raw_stream(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.rdx_merge_topk(dev.index or 0, p(part_score), p(part_row), p(part_count), int(P), int(nq), int(k),
                               p(out_score), p(out_row), p(out_count), L.RDX_DEVICE, ctypes.c_void_p(stream)))


def merge_topk(part_score: np.ndarray, part_row: np.ndarray, part_count: np.ndarray, k: int, device: int = 0):
    lib = L.load(require_gpu=True)
    ps = np.ascontiguousarray(part_score, dtype=np.float32)
    pr = np.ascontiguousarray(part_row, dtype=np.int64)
    pc = np.ascontiguousarray(part_count, dtype=np.int32)
    P, nq = pc.shape
    sc = np.empty((nq, k), dtype=np.float32)
    ro = np.empty((nq, k), dtype=np.int64)
    cn = np.empty((nq,), dtype=np.int32)
    L.check(lib.rdx_merge_topk(device, _np_ptr(ps), _np_ptr(pr), _np_ptr(pc), P, nq, int(k),
                               _np_ptr(sc), _np_ptr(ro), _np_ptr(cn), L.RDX_HOST, None))
    return sc, ro, cn
