"""MultiDeviceIndex — ONE engine object whose rows live in the HBM of several MI355X, driven from ONE process.

Why it exists. The reference serves from one process that holds one shared `collection` (reference app.py:42-67) which
the retriever calls synchronously (src/rag/retriever.py:215-220, 380-385). `rag_dpo_amd.sharded` scales the search the
torch.distributed way (one process per GPU, RCCL all-gather of the partials: bench.py); THIS class puts the
same row sharding behind the `collection` boundary, so that `Collection.query` spans every visible GPU without the
application knowing: `Collection(..., devices=[0, 1, ...])` or `RDX_DEVICES=all`.

Layout. Rows are dealt to the devices as they arrive (water-filling: a batch goes, cut into at most one contiguous piece
per device, to the devices holding the fewest rows), each device's shard is a plain `HipIndex` with its own stream, and
every shard answers with COLLECTION row ids through its row-id map (`rdx_index_set_row_ids`: local row -> global row,
strictly increasing inside a shard, so tie order inside a shard is the global tie order). A search is: the query batch
to every device (each shard's own H2D copy of B*d*4 bytes), the D scans concurrently (one host thread per device; ctypes
releases the GIL and every call blocks on its own stream only), then ONE merge of the D partial top-k lists on the first
device with the library's merge kernel (`rdx_merge_topk`, the same kernel the multi-process path runs after its
all-gather). Scores are computed by the same fixed-order arithmetic wherever a row lives, so the result is bit-identical
to a single-device index holding all rows (tests/test_multi_device.py; on one GPU the "devices" may repeat, e.g. [0, 0, 0]).

No arithmetic happens in this file: it moves rows and ids between the host layer and the per-device shards.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np


def _hip_shard(dim: int, device: int):
    from .engine import HipIndex
    return HipIndex(dim, device)


def _hip_merge(part_score, part_row, part_count, k: int, device: int):
    from .engine import merge_topk
    return merge_topk(part_score, part_row, part_count, k, device)


class _Ids:
    """append-only int64 array with amortised growth (a shard's local row -> collection row map on the host)"""

    def __init__(self, a=None):
        self.n = 0 if a is None else int(a.shape[0])
        self.buf = np.zeros(max(1024, self.n), dtype=np.int64)
        if a is not None:
            self.buf[: self.n] = a

    def append(self, a: np.ndarray):
        need = self.n + a.shape[0]
        if need > self.buf.shape[0]:
            b = np.zeros(max(need, self.buf.shape[0] * 2), dtype=np.int64)
            b[: self.n] = self.buf[: self.n]
            self.buf = b
        self.buf[self.n: need] = a
        self.n = need

    @property
    def a(self) -> np.ndarray:
        return self.buf[: self.n]


class MultiMask:
    """one resident bitmap per shard"""

    def __init__(self, parts, rows: int):
        self.parts, self.rows = parts, rows

    def close(self):
        for p in self.parts:
            if p is not None and hasattr(p, "close"):
                p.close()
        self.parts = []


class MultiDeviceIndex:
    def __init__(self, dim: int, devices: Sequence[int], shard_factory: Optional[Callable] = None,
                 merge: Optional[Callable] = None):
        if not devices:
            raise ValueError("MultiDeviceIndex needs at least one device")
        self.dim = int(dim)
        self.devices = [int(d) for d in devices]
        self.device = self.devices[0]
        self._shards = [(shard_factory or _hip_shard)(self.dim, d) for d in self.devices]
        self._merge = merge or _hip_merge
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="rdx-dev")
        D = len(self.devices)
        self._glob: List[_Ids] = [_Ids() for _ in range(D)]   # per shard: local row -> global row
        self._dev_of = np.zeros(0, dtype=np.int16)     # global row -> shard
        self._loc_of = np.zeros(0, dtype=np.int64)     # global row -> local row in that shard
        self._n = 0

    # ---- helpers ----------------------------------------------------------------------------
    def __len__(self) -> int:
        return self._n

    def _each(self, fn, items=None):
        """fn(shard index) on every shard, concurrently; exceptions propagate (first one wins)"""
        idx = list(range(len(self._shards))) if items is None else items
        if len(idx) == 1:
            return [fn(idx[0])]
        futs = [self._pool.submit(fn, i) for i in idx]
        out, err = [], None
        for f in futs:                                    # wait for ALL of them before reporting a failure
            try:
                out.append(f.result())
            except Exception as e:                        # noqa: BLE001 - re-raised below
                err = err or e
                out.append(None)
        if err is not None:
            raise err
        return out

    def _plan(self, n: int) -> List[int]:
        """rows of an n-row batch per shard (contiguous pieces in shard order): fill the emptiest shards first"""
        D = len(self._shards)
        have = np.array([g.n for g in self._glob], dtype=np.int64)
        level = -(-(int(have.sum()) + n) // D)            # ceil: the common fill level after the batch
        take = np.clip(level - have, 0, None)
        out, left = [], n
        for d in range(D):
            t = int(min(take[d], left))
            out.append(t)
            left -= t
        if left:                                          # rounding: whatever is left goes to the emptiest shard
            out[int(np.argmin(have + np.array(out)))] += left
        return out

    def _grow_maps(self, n: int):
        if self._dev_of.shape[0] < n:
            cap = max(n, self._dev_of.shape[0] * 3 // 2 + 1024)
            for name in ("_dev_of", "_loc_of"):
                a = getattr(self, name)
                b = np.zeros(cap, dtype=a.dtype)
                b[: a.shape[0]] = a
                setattr(self, name, b)

    def _as_rows(self, rows):
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and isinstance(rows, torch.Tensor):
            return rows                                   # sliced per shard below; HipIndex takes device or host tensors
        a = np.ascontiguousarray(rows, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError(f"expected [n][{self.dim}] embeddings, got shape {a.shape}")
        return a

    # ---- ingest -----------------------------------------------------------------------------
    def _add(self, rows, stored: bool):
        rows = self._as_rows(rows)
        n = int(rows.shape[0])
        if n == 0:
            return
        plan = self._plan(n)
        g0 = self._n
        pieces, off = [], 0
        for d, t in enumerate(plan):
            if t:
                pieces.append((d, off, t))
                off += t

        def put(j):
            d, o, t = pieces[j]
            sh = self._shards[d]
            piece = rows[o:o + t]
            if hasattr(piece, "is_cuda"):                 # torch tensor: the shard wants it on ITS device or on the host
                piece = piece.to(f"cuda:{self.devices[d]}") if piece.is_cuda else piece
            (sh.add_stored if stored else sh.add)(piece)

        try:
            self._each(put, list(range(len(pieces))))
        except Exception:
            # a rejected batch (NaN/Inf) must store nothing anywhere: roll the shards that took their piece back
            for d, o, t in pieces:
                have = self._glob[d].n
                if len(self._shards[d]) > have:
                    self._shards[d].compact(np.arange(have, dtype=np.int64))
                    if have:
                        self._shards[d].set_row_ids(0, self._glob[d].a)
            raise
        self._grow_maps(g0 + n)
        for d, o, t in pieces:
            l0 = self._glob[d].n
            ids = np.arange(g0 + o, g0 + o + t, dtype=np.int64)
            self._shards[d].set_row_ids(l0, ids)
            self._glob[d].append(ids)
            self._dev_of[g0 + o: g0 + o + t] = d
            self._loc_of[g0 + o: g0 + o + t] = np.arange(l0, l0 + t, dtype=np.int64)
        self._n = g0 + n

    def add(self, rows):
        self._add(rows, stored=False)

    def add_stored(self, rows):
        self._add(rows, stored=True)

    def reserve(self, rows: int):
        per = -(-int(rows) // len(self._shards))
        self._each(lambda d: self._shards[d].reserve(per))

    def set_option(self, name: str, value: int):
        if name == "row_base":
            raise ValueError("row_base is owned by the multi-device layer (row id maps)")
        self._each(lambda d: self._shards[d].set_option(name, value))

    def _split(self, row_ids):
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        if ids.size and (ids.min() < 0 or ids.max() >= self._n):
            raise ValueError(f"row id out of range [0, {self._n})")
        dev = self._dev_of[ids]
        return ids, [np.flatnonzero(dev == d) for d in range(len(self._shards))]

    def update(self, row_ids, rows):
        ids, where = self._split(row_ids)
        a = np.ascontiguousarray(rows, dtype=np.float32)
        if a.ndim != 2 or a.shape != (ids.shape[0], self.dim):
            raise ValueError("update: rows must be [len(row_ids)][dim]")
        if not np.isfinite(a).all():                      # checked up front: an update must not land on some shards only
            raise ValueError("embeddings contain NaN or Inf")
        self._each(lambda d: self._shards[d].update(self._loc_of[ids[where[d]]], a[where[d]]) if where[d].size else None)

    def get(self, row_ids) -> np.ndarray:
        ids, where = self._split(row_ids)
        out = np.empty((ids.shape[0], self.dim), dtype=np.float32)

        def one(d):
            if where[d].size:
                out[where[d]] = self._shards[d].get(self._loc_of[ids[where[d]]])
        self._each(one)
        return out

    def compact(self, keep_rows):
        keep = np.ascontiguousarray(keep_rows, dtype=np.int64)
        if keep.size and (keep[0] < 0 or keep[-1] >= self._n or (np.diff(keep) <= 0).any()):
            raise ValueError("compact: keep list must be strictly ascending row ids")
        dev, loc = self._dev_of[keep], self._loc_of[keep]
        new_glob = []
        for d in range(len(self._shards)):
            sel = np.flatnonzero(dev == d)                # new global ids (= ranks in keep) of the rows shard d keeps
            new_glob.append((sel.astype(np.int64), loc[sel]))

        def one(d):
            ids, locs = new_glob[d]
            self._shards[d].compact(locs)                 # ascending: loc is increasing in the global row inside a shard
            if ids.size:
                self._shards[d].set_row_ids(0, ids)
        self._each(one)
        self._n = int(keep.shape[0])
        self._dev_of = dev.astype(np.int16).copy()
        self._loc_of = np.empty(self._n, dtype=np.int64)
        for d, (ids, _) in enumerate(new_glob):
            self._glob[d] = _Ids(ids)
            self._loc_of[ids] = np.arange(ids.shape[0], dtype=np.int64)

    # ---- search -----------------------------------------------------------------------------
    def _shard_bits(self, allow_bits: np.ndarray):
        """collection bitmap -> one bitmap per shard (bit l of shard d = bit glob[d][l] of the collection's)"""
        words = np.ascontiguousarray(allow_bits, dtype=np.uint32)
        if words.shape[0] != (self._n + 31) // 32:
            raise ValueError("allow_bits must hold ceil(count/32) words")
        m = np.unpackbits(words.view(np.uint8), bitorder="little")[: self._n]
        out = []
        for g in self._glob:
            mm = m[g.a]
            pad = np.zeros((mm.shape[0] + 31) // 32 * 32, dtype=np.uint8)
            pad[: mm.shape[0]] = mm
            out.append(np.packbits(pad.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy())
        return out

    def make_mask(self, allow_bits: np.ndarray) -> MultiMask:
        bits = self._shard_bits(allow_bits)
        parts = self._each(lambda d: self._shards[d].make_mask(bits[d]) if hasattr(self._shards[d], "make_mask") else bits[d])
        return MultiMask(parts, self._n)

    def search(self, queries, k: int, allow_bits: Optional[np.ndarray] = None, mask: Optional[MultiMask] = None):
        """Host in / host out, like HipIndex.search: (score f32[nq,k], row i64[nq,k], count i32[nq]) with COLLECTION row ids"""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq][{self.dim}] query embeddings, got shape {q.shape}")
        if mask is not None and allow_bits is not None:
            raise ValueError("pass allow_bits or mask, not both")
        if mask is not None and mask.rows != self._n:
            raise ValueError("the mask was made for another state of the rows (a mask does not outlive a write)")
        bits = self._shard_bits(allow_bits) if allow_bits is not None else None
        live = [d for d in range(len(self._shards)) if self._glob[d].n > 0]
        nq = q.shape[0]
        if not live:
            return (np.full((nq, k), -np.inf, np.float32), np.full((nq, k), -1, np.int64), np.zeros(nq, np.int32))

        def one(d):
            sh = self._shards[d]
            if mask is not None:
                part = mask.parts[d]
                if isinstance(part, np.ndarray):
                    return sh.search(q, k, part)
                return sh.search(q, k, mask=part)
            return sh.search(q, k, bits[d]) if bits is not None else sh.search(q, k)
        parts = self._each(one, live)
        if len(parts) == 1:
            return parts[0]
        if k == 0:
            return parts[0][0], parts[0][1], np.zeros(nq, np.int32)
        # ONE merge call whatever D and k: the library folds the parts pairwise when D * k exceeds what its merge kernel ranks at once
        return self._merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]),
                           int(k), self.device)

    def search_device(self, queries, k: int, out_score=None, out_row=None, out_count=None, mask: Optional[MultiMask] = None):
        """Device in / device out (a batch caller behind `Collection(devices=...)`: nothing crosses PCIe). `queries`: [nq][dim]
        fp32 torch tensor on the FIRST device, produced on its current stream; the results land on the first device too, enqueued on
        its current stream: (score f32[nq,k], row i64[nq,k] COLLECTION row ids, count i32[nq]).
        The batch goes to every other device by a peer copy, every shard's search is ENQUEUED (rdx_search_async: the scans of the D
        devices run side by side), the host then completes each of them (fallback passes, if any), the partials travel to the
        first device by peer copies and ONE device merge ranks them. Same arithmetic wherever a row lives: bit-identical to search()."""
        import torch
        from .engine import merge_topk_device
        if k < 1:
            raise ValueError("search_device needs k >= 1")
        if mask is not None and mask.rows != self._n:
            raise ValueError("the mask was made for another state of the rows (a mask does not outlive a write)")
        dev0 = torch.device("cuda", self.devices[0])
        if not (queries.is_cuda and queries.device == dev0 and queries.dtype == torch.float32 and queries.dim() == 2
                and queries.shape[1] == self.dim and queries.is_contiguous()):
            raise ValueError(f"expected a contiguous [nq][{self.dim}] float32 tensor on {dev0}")
        nq = int(queries.shape[0])
        if nq > 4096:
            raise ValueError("search_device takes at most 4096 queries per call")
        if out_score is None:
            out_score = torch.empty((nq, k), dtype=torch.float32, device=dev0)
            out_row = torch.empty((nq, k), dtype=torch.int64, device=dev0)
            out_count = torch.empty((nq,), dtype=torch.int32, device=dev0)
        live = [d for d in range(len(self._shards)) if self._glob[d].n > 0]
        if not live:
            out_score.fill_(float("-inf")); out_row.fill_(-1); out_count.zero_()
            return out_score, out_row, out_count
        key = (nq, int(k), len(live))
        bufs = getattr(self, "_dev_bufs", None)
        if bufs is None or bufs[0] != key:
            per = {}
            for d in live:
                dd = torch.device("cuda", self.devices[d])
                per[d] = (torch.empty((nq, self.dim), dtype=torch.float32, device=dd) if dd != dev0 else None,
                          torch.empty((nq, k), dtype=torch.float32, device=dd), torch.empty((nq, k), dtype=torch.int64, device=dd),
                          torch.empty((nq,), dtype=torch.int32, device=dd))
            gathered = (torch.empty((len(live), nq, k), dtype=torch.float32, device=dev0),
                        torch.empty((len(live), nq, k), dtype=torch.int64, device=dev0),
                        torch.empty((len(live), nq), dtype=torch.int32, device=dev0))
            bufs = self._dev_bufs = (key, per, gathered)
        _, per, (g_s, g_r, g_c) = bufs
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev0))            # the batch exists on the first device from here on
        for d in live:
            dd = torch.device("cuda", self.devices[d])
            q_d, s_d, r_d, c_d = per[d]
            with torch.cuda.device(dd):
                st = torch.cuda.current_stream(dd)
                if q_d is not None:
                    st.wait_event(ready)
                    q_d.copy_(queries, non_blocking=True)         # peer copy over xGMI, on the receiving device's stream
                else:
                    q_d = queries
                self._shards[d].search_device_async(q_d, k, s_d, r_d, c_d, None, mask.parts[d] if mask is not None else None)
        for d in live:                                            # host halves: every shard's search complete (results final)
            self._shards[d].search_wait()
        if len(live) == 1:
            _, s_d, r_d, c_d = per[live[0]]
            out_score.copy_(s_d, non_blocking=True); out_row.copy_(r_d, non_blocking=True); out_count.copy_(c_d, non_blocking=True)
            return out_score, out_row, out_count
        s0 = torch.cuda.current_stream(dev0)
        for i, d in enumerate(live):
            dd = torch.device("cuda", self.devices[d])
            _, s_d, r_d, c_d = per[d]
            with torch.cuda.device(dd):
                st = torch.cuda.current_stream(dd)
                g_s[i].copy_(s_d, non_blocking=True); g_r[i].copy_(r_d, non_blocking=True); g_c[i].copy_(c_d, non_blocking=True)
                if st != s0:
                    done = torch.cuda.Event()
                    done.record(st)
                    s0.wait_event(done)
        with torch.cuda.device(dev0):
            merge_topk_device(g_s, g_r, g_c, int(k), out_score, out_row, out_count)
        return out_score, out_row, out_count

    def last_stats(self) -> dict:
        """per-shard stats of the last search, plus the sums of the additive counters"""
        per = [sh.last_stats() for sh in self._shards if hasattr(sh, "last_stats")]
        out = {"shards": per}
        for key in ("emitted", "rescored", "exact_queries", "retried_queries", "sample_rows"):
            out[key] = sum(p.get(key, 0) for p in per)
        out["path"] = max((p.get("path", 0) for p in per), default=0)
        return out

    def close(self):
        for sh in self._shards:
            if hasattr(sh, "close"):
                sh.close()
        self._pool.shutdown(wait=False)


def multi_device_factory(devices: Sequence[int]):
    """engine_factory for Collection / PersistentClient: rows over `devices` (one process, one collection object)"""
    devs = [int(d) for d in devices]

    def factory(dim: int, device: int = 0):
        if len(devs) == 1:
            return _hip_shard(dim, devs[0])
        return MultiDeviceIndex(dim, devs)
    return factory


def visible_devices() -> List[int]:
    from . import _lib as L
    import ctypes
    lib = L.load(require_gpu=True)
    n = ctypes.c_int(0)
    L.check(lib.rdx_device_count(ctypes.byref(n)))
    return list(range(n.value))
