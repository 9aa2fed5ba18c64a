"""Chroma-shaped `Collection` / `PersistentClient` whose vectors live in MI355X HBM (librdx).

Mirrors exactly the part of chromadb's API that RAG-DPO calls:
    collection.query(query_embeddings=, n_results=, where=, include=)   src/rag/retriever.py:215-220, 380-385
    collection.add(ids=, documents=, embeddings=, metadatas=)           src/processing/create_chromadb_index.py:374-379
    collection.get(limit=, offset=, where=, ids=, include=)             src/rag/bm25_index.py:211-215 and others
    collection.count() / delete(ids=) / update(ids=, metadatas=)        ingest_enterprise.py:272, tag_all_chunks.py:215
    client.get_collection / create_collection / delete_collection       app.py:58-59, create_chromadb_index.py:93-130
Everything that crosses this boundary is plain Python lists/dicts/strs/floats, as with chromadb.
ids/documents/metadata stay on the host; embeddings go to the device index (`engine`). The engine is
always librdx (rag_dpo_amd.engine.HipIndex): there is no CPU search path in this package.

Distances keep Chroma's cosine convention: distance = 1 - cos, ascending (so that
`similarity_score = 1/(1+distance)` reference retriever.py:39-42 and the 0.80 relevance threshold
reference validators.py:64-81 keep their meaning). The search itself is exact (brute force), not HNSW.
"""
from __future__ import annotations

import json
import os
import threading
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from . import where as W

DEFAULT_INCLUDE_QUERY = ["metadatas", "documents", "distances"]
DEFAULT_INCLUDE_GET = ["metadatas", "documents"]
_VALID_INCLUDE = {"embeddings", "documents", "metadatas", "distances", "uris", "data"}


class DuplicateIDError(ValueError):
    pass


class NotFoundError(ValueError):
    pass


def _default_engine_factory(dim: int, device: int):
    from .engine import HipIndex   # raises RdxUnavailable when librdx / a gfx950 GPU is missing
    return HipIndex(dim, device)


def _devices_from_env() -> Optional[List[int]]:
    """RDX_DEVICES=all | "0,1,2,3": spread every collection of this process over these GPUs (rag_dpo_amd.multi_device)"""
    v = os.environ.get("RDX_DEVICES", "").strip()
    if not v:
        return None
    if v.lower() == "all":
        from .multi_device import visible_devices
        return visible_devices()
    return [int(x) for x in v.split(",") if x.strip() != ""]


def _fsync_dir(path: str):
    """make a rename / file creation in `path` durable (the directory entry itself)"""
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return
    try:
        os.fsync(fd)
    except OSError:
        pass
    finally:
        os.close(fd)


def _is_device_tensor(x) -> bool:
    import sys
    torch = sys.modules.get("torch")
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


def _as_matrix(embeddings, what: str):
    """-> contiguous fp32 [n][dim]: a numpy array, or the caller's torch CUDA tensor passed through untouched
    (EmbeddingProvider.embed_device -> add: the batch never becomes Python floats)"""
    if _is_device_tensor(embeddings):
        import torch
        t = embeddings if embeddings.dtype == torch.float32 else embeddings.float()
        if t.dim() != 2 or t.shape[0] == 0 or t.shape[1] == 0:
            raise ValueError(f"Expected {what} to be a non-empty [n][dim] tensor, got shape {tuple(t.shape)}")
        return t.contiguous()
    a = np.asarray(embeddings, dtype=np.float32)
    if a.ndim == 1 and a.size > 0:
        a = a[None, :]
    if a.ndim != 2 or a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError(f"Expected {what} to be a non-empty list of embeddings, got shape {a.shape}")
    return np.ascontiguousarray(a)


class Collection:
    def __init__(self, name: str, metadata: Optional[dict] = None, device: int = 0,
                 engine_factory: Optional[Callable[[int, int], Any]] = None, _client=None,
                 devices: Optional[Sequence[int]] = None):
        """devices=[0, 1, ...] (or RDX_DEVICES in the environment): the rows of this ONE collection object are sharded over
        these GPUs and every query spans them (rag_dpo_amd.multi_device) — the reference's single shared `collection`
        (app.py:42-67) on up to 8 MI355X. Default: one GPU (`device`)."""
        self.name = name
        self.metadata = dict(metadata or {})
        space = self.metadata.get("hnsw:space", "cosine")
        if space != "cosine":
            raise ValueError(f"only the cosine space is implemented (the reference uses 'hnsw:space': 'cosine'), got {space!r}")
        self._device = device
        if engine_factory is None:
            devs = list(devices) if devices is not None else _devices_from_env()
            if devs is not None and len(devs) > 1:
                from .multi_device import multi_device_factory
                engine_factory = multi_device_factory(devs)
            elif devs:
                self._device = device = int(devs[0])
        self._factory = engine_factory or _default_engine_factory
        self._engine = None
        self._dim: Optional[int] = None
        self._ids: List[str] = []              # row -> id ("" when tombstoned)
        self._docs: List[Optional[str]] = []
        self._row_of: Dict[str, int] = {}
        self._alive = np.zeros(0, dtype=bool)
        self._n_dead = 0
        self._cols: Dict[str, W.Column] = {}
        self._lock = threading.RLock()
        self._client = _client
        self._meta_cache: Dict[int, Optional[dict]] = {}   # row -> metadata dict as last assembled (dropped on any write)
        # `where` filters are a handful of fixed shapes re-sent with every question (reference src/rag/pipeline.py:35-71,
        # pages/1_Chat.py:245-247): canonical JSON of the filter -> (packed bitmap, the engine's HBM-resident copy or None).
        # Dropped on every write (add / update / delete / compaction): a bitmap describes one state of the rows.
        self._mask_cache: "Dict[str, tuple]" = {}
        self.mask_cache_hits = 0
        self._dir: Optional[str] = None        # set by PersistentClient: where the snapshot + journal live
        self._replaying = False

    # ---- small helpers ----------------------------------------------------------------------
    @property
    def _rows(self) -> int:
        return len(self._ids)

    def _ensure_engine(self, dim: int):
        if self._engine is None:
            self._engine = self._factory(dim, self._device)
            self._dim = dim
        elif dim != self._dim:
            raise ValueError(f"Embedding dimension {dim} does not match collection dimensionality {self._dim}")

    def _set_meta(self, row: int, meta: Optional[dict], replace: bool):
        self._meta_cache.pop(row, None)
        if self._mask_cache:
            self._drop_masks()
        if replace:
            for col in self._cols.values():
                col.kind[row] = W.K_MISSING
        if not meta:
            return
        for k, v in meta.items():
            if not isinstance(k, str):
                raise ValueError(f"Expected metadata key to be a str, got {k!r}")
            col = self._cols.get(k)
            if col is None:
                col = self._cols[k] = W.Column(self._rows)
            col.set(row, v)

    def _meta_of(self, row: int) -> Optional[dict]:
        out = {}
        for k, col in self._cols.items():
            v = col.get(row)
            if v is not None:
                out[k] = v
        return out or None

    _META_CACHE_MAX = 200_000

    def _metas_of(self, rows) -> List[Optional[dict]]:
        """metadata dicts of the result rows. Rows seen before come out of a per-row cache (a RAG store answers with the
        same popular chunks again and again; rebuilding 50 dicts of 18 fields from the columns costs more than the GPU
        search at the reference's corpus size); the caller gets its own copies, as with chromadb."""
        rows = [int(r) for r in rows]
        cache = self._meta_cache
        miss = [r for r in rows if r not in cache]
        if miss:
            if len(cache) + len(miss) > self._META_CACHE_MAX:
                cache.clear()
            for r, m in zip(miss, self._metas_from_columns(miss)):
                cache[r] = m
        return [dict(m) if m else None for m in (cache[r] for r in rows)]

    def _metas_from_columns(self, rows) -> List[Optional[dict]]:
        """one vectorised gather per column instead of one Column.get per (row, key)"""
        rr = np.asarray(rows, dtype=np.int64)
        out: List[dict] = [{} for _ in range(rr.shape[0])]
        if rr.shape[0] == 0:
            return []
        for key, col in self._cols.items():
            kinds = col.kind[rr]
            if not kinds.any():
                continue
            vocab = col.vocab
            k0 = int(kinds[0])
            if k0 != W.K_MISSING and (kinds == k0).all():     # the usual case: one kind for the whole column
                if k0 == W.K_STR:
                    vals = [vocab[c] for c in col.code[rr].tolist()]
                elif k0 == W.K_INT:
                    vals = col.num[rr].astype(np.int64).tolist()
                elif k0 == W.K_FLOAT:
                    vals = col.num[rr].tolist()
                else:
                    vals = (col.num[rr] != 0).tolist()
                for d, v in zip(out, vals):
                    d[key] = v
                continue
            kl = kinds.tolist()
            nums = col.num[rr].tolist()
            codes = col.code[rr].tolist()
            for j, kd in enumerate(kl):
                if kd == W.K_STR:
                    out[j][key] = vocab[codes[j]]
                elif kd == W.K_INT:
                    out[j][key] = int(nums[j])
                elif kd == W.K_FLOAT:
                    out[j][key] = nums[j]
                elif kd == W.K_BOOL:
                    out[j][key] = bool(nums[j])
        return [d or None for d in out]

    def _grow_cols(self, n: int):
        for col in self._cols.values():
            col.resize(n)
        if self._alive.shape[0] < n:
            a = np.zeros(max(n, self._alive.shape[0] * 3 // 2 + 16), dtype=bool)
            a[: self._alive.shape[0]] = self._alive
            self._alive = a

    def _mask(self, where: Optional[dict]) -> Optional[np.ndarray]:
        """row bitmap source: `where` pre-filter AND not-deleted; None = everything passes"""
        n = self._rows
        m = W.evaluate(where, self._cols, n)
        if self._n_dead:
            m = self._alive[:n].copy() if m is None else (m & self._alive[:n])
        return m

    _MASK_CACHE_MAX = 32

    def _drop_masks(self):
        for _, res in self._mask_cache.values():
            if res is not None and hasattr(res, "close"):
                res.close()
        self._mask_cache.clear()

    def _search_args(self, where: Optional[dict]) -> dict:
        """-> keyword arguments for engine.search: nothing (no filter, no tombstones), mask= (resident bitmap of a filter
        seen before or just uploaded) or allow_bits= (engines without resident masks)"""
        if where in (None, {}) and not self._n_dead:
            return {}
        try:
            key = json.dumps(where, sort_keys=True, ensure_ascii=False, allow_nan=False)
        except (TypeError, ValueError):
            key = None                                  # not canonicalisable: W.evaluate will say what is wrong with it
        ent = self._mask_cache.get(key) if key is not None else None
        if ent is None:
            m = self._mask(where)
            if m is None:
                return {}
            bits = W.pack_bits(m)
            res = self._engine.make_mask(bits) if hasattr(self._engine, "make_mask") else None
            ent = (bits, res)
            if key is not None:
                if len(self._mask_cache) >= self._MASK_CACHE_MAX:
                    old = next(iter(self._mask_cache))          # oldest entry (insertion order)
                    _, r0 = self._mask_cache.pop(old)
                    if r0 is not None and hasattr(r0, "close"):
                        r0.close()
                self._mask_cache[key] = ent
        else:
            self.mask_cache_hits += 1
        return {"mask": ent[1]} if ent[1] is not None else {"allow_bits": ent[0]}

    @staticmethod
    def _check_include(include: Sequence[str], allowed: set):
        for inc in include:
            if inc not in _VALID_INCLUDE or inc not in allowed:
                raise ValueError(f"Expected include item to be one of {sorted(allowed)}, got {inc}")

    def _maybe_compact(self):
        if self._n_dead > max(1024, self._rows // 5):
            self._compact()

    def _compact(self):
        n = self._rows
        keep = np.flatnonzero(self._alive[:n])
        self._engine.compact(keep)
        self._ids = [self._ids[i] for i in keep]
        self._docs = [self._docs[i] for i in keep]
        self._cols = {k: c.take(keep) for k, c in self._cols.items()}
        self._alive = np.ones(len(keep), dtype=bool)
        self._row_of = {s: i for i, s in enumerate(self._ids)}
        self._n_dead = 0
        self._meta_cache.clear()
        self._drop_masks()

    # ---- chromadb.Collection API ---------------------------------------------------------------
    def count(self) -> int:
        with self._lock:
            return self._rows - self._n_dead

    def add(self, ids, embeddings=None, metadatas=None, documents=None, _stored: bool = False, **_ignored):
        """reference create_chromadb_index.py:374-379, ingest_enterprise.py:241-246. Existing ids are skipped
        (chromadb's add never overwrites); duplicate ids inside one call raise DuplicateIDError."""
        if isinstance(ids, str):
            ids = [ids]
        ids = list(ids)
        if not ids:
            raise ValueError("Expected IDs to be a non-empty list, got 0 IDs")
        if embeddings is None:
            raise ValueError("this collection has no embedding function: pass embeddings= "
                             "(the reference always does, create_chromadb_index.py:374-379)")
        emb = _as_matrix(embeddings, "embeddings")
        n = len(ids)
        if emb.shape[0] != n:
            raise ValueError(f"Unequal lengths for fields: ids: {n}, embeddings: {emb.shape[0]}")
        for name, lst in (("metadatas", metadatas), ("documents", documents)):
            if lst is not None and len(lst) != n:
                raise ValueError(f"Unequal lengths for fields: ids: {n}, {name}: {len(lst)}")
        if len(set(ids)) != n:
            seen, dups = set(), set()
            for s in ids:
                (dups if s in seen else seen).add(s)
            raise DuplicateIDError(f"Expected IDs to be unique, found duplicates of: {', '.join(sorted(dups))}")
        for s in ids:
            if not isinstance(s, str) or not s:
                raise ValueError(f"Expected ID to be a non-empty str, got {s!r}")
        with self._lock:
            self._ensure_engine(emb.shape[1])
            fresh = [i for i, s in enumerate(ids) if s not in self._row_of]
            if not fresh:
                return
            if metadatas is not None:   # validate EVERYTHING before anything is stored (host and device rows stay in step)
                for i in fresh:
                    W.check_meta(metadatas[i])
            self._drop_masks()
            sel = emb if len(fresh) == n else (emb[fresh].contiguous() if _is_device_tensor(emb) else np.ascontiguousarray(emb[fresh]))
            row0 = self._rows
            if _stored:             # snapshot reload: rows are the stored (already normalised) values, kept verbatim
                self._engine.add_stored(sel)
            else:
                self._engine.add(sel)   # raises ValueError on NaN/Inf: nothing stored
            self._grow_cols(row0 + len(fresh))
            for j, i in enumerate(fresh):
                row = row0 + j
                self._ids.append(ids[i])
                self._docs.append(documents[i] if documents is not None else None)
                self._row_of[ids[i]] = row
                self._alive[row] = True
                self._set_meta(row, metadatas[i] if metadatas is not None else None, replace=False)
            self._log({"op": "add", "ids": [ids[i] for i in fresh],
                       "documents": None if documents is None else [documents[i] for i in fresh],
                       "metadatas": None if metadatas is None else [metadatas[i] for i in fresh]}, sel)

    def update(self, ids, embeddings=None, metadatas=None, documents=None, **_ignored):
        """reference tag_all_chunks.py:215,224 (metadatas only). Metadata updates MERGE into the stored dict,
        as chromadb does; ids that do not exist are ignored."""
        if isinstance(ids, str):
            ids = [ids]
        ids = list(ids)
        n = len(ids)
        emb = _as_matrix(embeddings, "embeddings") if embeddings is not None else None
        if _is_device_tensor(emb):
            emb = emb.cpu().numpy()    # in-place row updates are rare and small: through the host path
        for name, lst in (("embeddings", emb), ("metadatas", metadatas), ("documents", documents)):
            if lst is not None and len(lst) != n:
                raise ValueError(f"Unequal lengths for fields: ids: {n}, {name}: {len(lst)}")
        if metadatas is not None:
            for m in metadatas:
                W.check_meta(m)
        with self._lock:
            hit = [(i, self._row_of[s]) for i, s in enumerate(ids) if s in self._row_of]
            if emb is not None and hit:
                self._ensure_engine(emb.shape[1])
                self._engine.update(np.array([r for _, r in hit], dtype=np.int64),
                                    np.ascontiguousarray(emb[[i for i, _ in hit]]))
            for i, row in hit:
                if metadatas is not None and metadatas[i]:
                    self._set_meta(row, metadatas[i], replace=False)
                if documents is not None:
                    self._docs[row] = documents[i]
            if hit:
                idx = [i for i, _ in hit]
                self._log({"op": "update", "ids": [ids[i] for i in idx],
                           "documents": None if documents is None else [documents[i] for i in idx],
                           "metadatas": None if metadatas is None else [metadatas[i] for i in idx]},
                          None if emb is None else np.ascontiguousarray(emb[idx]))

    def upsert(self, ids, embeddings=None, metadatas=None, documents=None, **_ignored):
        if isinstance(ids, str):
            ids = [ids]
        ids = list(ids)
        with self._lock:
            old = [i for i, s in enumerate(ids) if s in self._row_of]
            new = [i for i, s in enumerate(ids) if s not in self._row_of]
            pick = lambda lst, idx: None if lst is None else [lst[i] for i in idx]
            emb = _as_matrix(embeddings, "embeddings") if embeddings is not None else None
            if _is_device_tensor(emb) and old:
                emb = emb.cpu().numpy()
            if old:
                self.update([ids[i] for i in old], None if emb is None else emb[old], pick(metadatas, old), pick(documents, old))
            if new:
                self.add([ids[i] for i in new], None if emb is None else emb[new], pick(metadatas, new), pick(documents, new))

    def delete(self, ids=None, where=None, **_ignored):
        """reference ingest_enterprise.py:272,304 (ids in batches of 5000). Rows are tombstoned (excluded from
        every search through the row bitmap) and compacted out of HBM once a fifth of the rows is dead."""
        with self._lock:
            rows: List[int] = []
            if ids is not None:
                if isinstance(ids, str):
                    ids = [ids]
                rows = [self._row_of[s] for s in ids if s in self._row_of]
                if where is not None and rows:
                    m = W.evaluate(where, self._cols, self._rows)
                    rows = [r for r in rows if m is None or m[r]]
            elif where is not None:
                m = self._mask(where)
                rows = np.flatnonzero(m).tolist() if m is not None else list(range(self._rows))
            else:
                raise ValueError("delete needs ids= or where=")
            gone = [self._ids[r] for r in rows if self._alive[r]]
            if gone:
                self._drop_masks()
                self._log({"op": "delete", "ids": gone}, None)
            for r in rows:
                if self._alive[r]:
                    self._alive[r] = False
                    self._meta_cache.pop(r, None)
                    self._n_dead += 1
                    del self._row_of[self._ids[r]]
                    self._ids[r] = ""
                    self._docs[r] = None
            if rows:
                self._maybe_compact()

    def get(self, ids=None, where=None, limit=None, offset=None, where_document=None, include=None, **_ignored):
        """reference bm25_index.py:211-215 (paging with limit/offset), create_chromadb_index.py:118,452-468,
        ingest_enterprise.py:142-145. Order = insertion order, like chromadb."""
        include = DEFAULT_INCLUDE_GET if include is None else list(include)
        self._check_include(include, {"embeddings", "documents", "metadatas", "uris", "data"})
        if where_document is not None:
            raise ValueError("where_document is not implemented (the reference never passes it)")
        with self._lock:
            if ids is not None:
                if isinstance(ids, str):
                    ids = [ids]
                rows = sorted(self._row_of[s] for s in set(ids) if s in self._row_of)
                if where is not None and rows:
                    m = W.evaluate(where, self._cols, self._rows)
                    rows = [r for r in rows if m is None or m[r]]
            else:
                m = self._mask(where)
                rows = list(range(self._rows)) if m is None else np.flatnonzero(m).tolist()
            off = int(offset or 0)
            rows = rows[off: off + int(limit)] if limit is not None else rows[off:]
            out = {
                "ids": [self._ids[r] for r in rows],
                "embeddings": None,
                "documents": [self._docs[r] for r in rows] if "documents" in include else None,
                "metadatas": self._metas_of(rows) if "metadatas" in include else None,
                "uris": None,
                "data": None,
                "included": include,
            }
            if "embeddings" in include:
                out["embeddings"] = (self._engine.get(np.array(rows, dtype=np.int64)) if rows and self._engine is not None
                                     else np.zeros((0, self._dim or 0), dtype=np.float32))
            return out

    def peek(self, limit: int = 10):
        return self.get(limit=limit, include=["embeddings", "metadatas", "documents"])

    def query(self, query_embeddings=None, n_results: int = 10, where=None, where_document=None, include=None,
              query_texts=None, **_ignored):
        """reference src/rag/retriever.py:215-220, 380-385 (one query, n_results=50, optional where, include
        documents+metadatas+distances) and create_chromadb_index.py:405-408 (no include/where)."""
        include = DEFAULT_INCLUDE_QUERY if include is None else list(include)
        self._check_include(include, _VALID_INCLUDE)
        if query_embeddings is None:
            raise ValueError("this collection has no embedding function: pass query_embeddings= "
                             "(the reference always does, retriever.py:215-220)")
        if where_document is not None:
            raise ValueError("where_document is not implemented (the reference never passes it)")
        if not isinstance(n_results, (int, np.integer)) or isinstance(n_results, bool) or n_results <= 0:
            raise ValueError(f"Number of requested results {n_results}, cannot be negative, or zero.")
        q = _as_matrix(query_embeddings, "query_embeddings")
        if _is_device_tensor(q):
            q = q.cpu().numpy()        # results are Python lists anyway; device callers use HipIndex.search_device
        nq = q.shape[0]
        with self._lock:
            if self._engine is None:   # nothing was ever added
                empty = [[] for _ in range(nq)]
                return {"ids": [[] for _ in range(nq)], "embeddings": None,
                        "documents": [list(e) for e in empty] if "documents" in include else None,
                        "metadatas": [list(e) for e in empty] if "metadatas" in include else None,
                        "distances": [list(e) for e in empty] if "distances" in include else None,
                        "uris": None, "data": None, "included": include}
            if q.shape[1] != self._dim:
                raise ValueError(f"Embedding dimension {q.shape[1]} does not match collection dimensionality {self._dim}")
            scores, rows, counts = self._engine.search(q, int(n_results), **self._search_args(where))
            dist = (np.float32(1.0) - scores).astype(np.float32)   # Chroma cosine distance, fp32 like chromadb
            out_ids, out_docs, out_meta, out_dist, out_emb = [], [], [], [], []
            for b in range(nq):
                rr = rows[b, : counts[b]].tolist()
                out_ids.append([self._ids[r] for r in rr])
                if "documents" in include:
                    out_docs.append([self._docs[r] for r in rr])
                if "metadatas" in include:
                    out_meta.append(self._metas_of(rr))
                if "distances" in include:
                    out_dist.append([float(x) for x in dist[b, : counts[b]]])
                if "embeddings" in include:
                    out_emb.append(self._engine.get(np.array(rr, dtype=np.int64)) if rr
                                   else np.zeros((0, self._dim), dtype=np.float32))
            return {
                "ids": out_ids,
                "embeddings": out_emb if "embeddings" in include else None,
                "documents": out_docs if "documents" in include else None,
                "metadatas": out_meta if "metadatas" in include else None,
                "distances": out_dist if "distances" in include else None,
                "uris": None,
                "data": None,
                "included": include,
            }

    def query_device(self, query_embeddings, n_results: int = 10, where=None):
        """Batch callers that already hold their query embeddings on the GPU (the provider's `embed_device`, reference
        src/utils/embedding_provider.py:139-145 feeding src/rag/retriever.py:215-220) and want the neighbours there too:
        `query_embeddings` a [nq][dim] fp32 torch CUDA tensor -> (distances f32[nq, n_results], rows i64[nq, n_results],
        counts i32[nq]) torch tensors on the collection's (first) device; `ids_of(rows)` maps rows to the Chroma ids. Same
        filter semantics and the same floats as query() — nothing crosses PCIe but the call itself."""
        import torch
        with self._lock:
            if self._engine is None or self._rows == 0:
                raise ValueError("query_device: the collection is empty")
            if not hasattr(self._engine, "search_device"):
                raise NotImplementedError("this collection's engine has no device-pointer search")
            q = query_embeddings.contiguous()
            nq, k = int(q.shape[0]), int(n_results)
            args = self._search_args(where)
            if "allow_bits" in args:
                raise NotImplementedError("query_device needs an engine with resident masks")
            dev = q.device
            s = torch.empty((nq, k), dtype=torch.float32, device=dev)
            r = torch.empty((nq, k), dtype=torch.int64, device=dev)
            c = torch.empty((nq,), dtype=torch.int32, device=dev)
            self._engine.search_device(q, k, s, r, c, mask=args.get("mask"))
            return (1.0 - s), r, c            # Chroma cosine distance, fp32 (padding entries: distance +inf, row -1)

    def ids_of(self, rows) -> List[List[str]]:
        """Chroma ids of row ids as returned by query_device (negative = padding, skipped)"""
        with self._lock:
            rr = rows.tolist() if hasattr(rows, "tolist") else rows
            return [[self._ids[x] for x in row if x >= 0] for row in rr]

    def modify(self, name: Optional[str] = None, metadata: Optional[dict] = None):
        with self._lock:
            if name is not None:
                if self._client is not None:
                    self._client._rename(self.name, name)
                self.name = name
            if metadata is not None:
                self.metadata = dict(metadata)
            if self._dir:
                self._write_header(self._dir, self._snap_rows)

    # ---- persistence: own shard format ------------------------------------------
    # <dir>/collection.json              name, metadata, dim, rows and GENERATION g of the current snapshot (format 3)
    # <dir>/snap<g>.embeddings.f32.npy   snapshot: the stored (normalised) fp32 rows, in row order; reloaded VERBATIM
    # <dir>/snap<g>.records.jsonl        snapshot: one {"id", "document", "metadata"} per row
    # <dir>/journal<g>.jsonl + .f32      every add/update/delete since snapshot g, appended (and fsync'ed) before the call
    #                                    returns: chromadb's PersistentClient is durable per call and the reference never
    #                                    calls persist() (create_chromadb_index.py writes, app.py reads in another process).
    # Crash consistency. A journal record is committed by its complete jsonl line, written AFTER its vectors; the line
    # carries the byte range of those vectors in journal.f32, so orphan floats of a writer killed between the two writes
    # are never read, and opening a store truncates both files to the last committed record (a torn last line, or floats
    # without a line, vanish). persist() writes snapshot g+1 under new names, then replaces collection.json atomically
    # (the commit point: it names the generation), then deletes generation g — a crash at any point leaves either g or
    # g+1 complete, and the files of the other one are removed at the next open.
    _snap_rows = 0
    _gen = 0

    def _names(self, gen: Optional[int]):
        pre = "" if gen is None else f"snap{gen}."
        jp = "journal" if gen is None else f"journal{gen}"
        return {"emb": pre + "embeddings.f32.npy", "rec": pre + "records.jsonl", "jl": jp + ".jsonl", "jf": jp + ".f32"}

    def _cur_names(self):
        return self._names(self._gen if self._snap_format >= 3 else None)   # format 2 stores (round 1) had one unnamed generation

    def _write_header(self, path: str, rows: int, fmt: Optional[int] = None, gen: Optional[int] = None):
        os.makedirs(path, exist_ok=True)
        tmp = os.path.join(path, "collection.json.tmp")
        with open(tmp, "w", encoding="utf-8") as f:
            hdr = {"name": self.name, "metadata": self.metadata, "dim": self._dim, "rows": rows,
                   "format": self._snap_format if fmt is None else fmt, "gen": self._gen if gen is None else gen}
            if self._engine is not None and hasattr(self._engine, "xcd_shares"):
                try:   # what the scan has learned about this GPU's XCDs travels with the store (speed only; include/rdx.h)
                    hdr["xcd_shares"] = [round(x, 5) for x in self._engine.xcd_shares()]
                except Exception:
                    pass
            json.dump(hdr, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(path, "collection.json"))
        _fsync_dir(path)

    def _log(self, rec: dict, emb):
        if not self._dir or self._replaying:
            return
        nm = self._cur_names()
        rec["n_emb"] = 0 if emb is None else int(emb.shape[0])
        rec["dim"] = None if emb is None else int(emb.shape[1])
        if _is_device_tensor(emb):
            emb = emb.cpu().numpy()
        if emb is not None:
            raw = np.ascontiguousarray(emb, dtype=np.float32).tobytes()
            with open(os.path.join(self._dir, nm["jf"]), "ab") as f:
                rec["f32_off"] = f.seek(0, os.SEEK_END)
                rec["f32_len"] = len(raw)
                f.write(raw)
                f.flush()
                os.fsync(f.fileno())
        with open(os.path.join(self._dir, nm["jl"]), "a", encoding="utf-8") as f:   # the complete line commits the op
            f.write(json.dumps(rec, ensure_ascii=False) + "\n")
            f.flush()
            os.fsync(f.fileno())

    def _save(self, path: str):
        with self._lock:
            if self._n_dead:
                self._compact()
            os.makedirs(path, exist_ok=True)
            old = self._cur_names()
            # The object moves to generation g+1 only AFTER the header naming it is on disk. Until then every add / update /
            # delete must keep going to journal<g>, which collection.json still names: a snapshot that fails half way (ENOSPC
            # in np.save, say) leaves the store exactly as it was, and the writes acknowledged afterwards are found again.
            new_gen = self._gen + 1
            nm = self._names(new_gen)
            n = self._rows
            try:
                if n:
                    emb = self._engine.get(np.arange(n, dtype=np.int64))   # the stored (normalised) fp32 rows
                    with open(os.path.join(path, nm["emb"]), "wb") as f:
                        np.save(f, emb)
                        f.flush()
                        os.fsync(f.fileno())
                with open(os.path.join(path, nm["rec"]), "w", encoding="utf-8") as f:
                    for r in range(n):
                        f.write(json.dumps({"id": self._ids[r], "document": self._docs[r], "metadata": self._meta_of(r)},
                                           ensure_ascii=False) + "\n")
                    f.flush()
                    os.fsync(f.fileno())
                _fsync_dir(path)                      # the new generation's files exist before the header can name them
                self._write_header(path, n, fmt=3, gen=new_gen)   # commit point: the header names generation g+1
            except BaseException:
                # os.replace(collection.json) IS the commit point: if the header on disk names generation g+1 (the replace
                # happened and something after it raised — the directory fsync, an interrupt), the new files are the store and
                # must stay; the object follows the header, so that later writes go to the journal the header names.
                committed = False
                try:
                    with open(os.path.join(path, "collection.json"), encoding="utf-8") as f:
                        committed = int(json.load(f).get("gen", -1)) == new_gen
                except (OSError, ValueError):
                    pass
                if committed:
                    self._snap_format, self._gen, self._snap_rows = 3, new_gen, n
                else:
                    for fn in (nm["emb"], nm["rec"], "collection.json.tmp"):   # partial files of the generation that never committed
                        try:
                            os.remove(os.path.join(path, fn))
                        except OSError:
                            pass
                raise
            self._snap_format, self._gen, self._snap_rows = 3, new_gen, n
            for fn in old.values():
                fp = os.path.join(path, fn)
                if os.path.exists(fp):
                    os.remove(fp)

    _snap_format = 3

    def _load(self, path: str):
        with open(os.path.join(path, "collection.json"), encoding="utf-8") as f:
            meta = json.load(f)
        self.metadata = meta.get("metadata") or {}
        n = int(meta.get("rows", 0))
        self._snap_rows = n
        self._snap_format = int(meta.get("format", 2))
        self._gen = int(meta.get("gen", 0))
        nm = self._cur_names()
        # files of another generation: a persist() that was killed before or after its commit point
        keep = set(nm.values()) | {"collection.json"}
        for fn in os.listdir(path):
            if fn not in keep and (fn.startswith("snap") or fn.startswith("journal") or fn == "collection.json.tmp"
                                   or (self._snap_format >= 3 and fn in self._names(None).values())):
                os.remove(os.path.join(path, fn))
        self._replaying = True
        try:
            if n:
                emb = np.load(os.path.join(path, nm["emb"]), mmap_mode="r", allow_pickle=False)
                ids, docs, metas = [], [], []
                with open(os.path.join(path, nm["rec"]), encoding="utf-8") as f:
                    for line in f:
                        rec = json.loads(line)
                        ids.append(rec["id"])
                        docs.append(rec.get("document"))
                        metas.append(rec.get("metadata"))
                step = 65536
                for a in range(0, n, step):
                    b = min(n, a + step)
                    self.add(ids=ids[a:b], embeddings=np.asarray(emb[a:b]), documents=docs[a:b], metadatas=metas[a:b],
                             _stored=True)   # verbatim: scores after a reload are bit-identical to those before
            jp, fp = os.path.join(path, nm["jl"]), os.path.join(path, nm["jf"])
            if os.path.exists(jp):
                with open(jp, "rb") as f:
                    blob = f.read()
                f32_size = os.path.getsize(fp) if os.path.exists(fp) else 0
                raw = np.memmap(fp, dtype=np.uint8, mode="r") if f32_size else None
                good, f32_end, pos, legacy_pos = 0, 0, 0, 0
                while pos < len(blob):
                    nl = blob.find(b"\n", pos)
                    if nl < 0:
                        break                      # torn last line of a killed writer: the op never committed
                    try:
                        rec = json.loads(blob[pos:nl].decode("utf-8"))
                    except ValueError:
                        break
                    emb = None
                    if rec.get("n_emb"):
                        cnt = rec["n_emb"] * rec["dim"] * 4
                        off = rec.get("f32_off", legacy_pos)      # format 2 journals carry no offsets: running position
                        if off + cnt > f32_size:
                            break                  # its vectors never reached the disk: not committed
                        emb = np.frombuffer(raw[off: off + cnt].tobytes(), dtype=np.float32).reshape(rec["n_emb"], rec["dim"])
                        legacy_pos = off + cnt
                        f32_end = max(f32_end, off + cnt)
                    if rec["op"] == "add":
                        self.add(ids=rec["ids"], embeddings=emb, documents=rec["documents"], metadatas=rec["metadatas"])
                    elif rec["op"] == "update":
                        self.update(ids=rec["ids"], embeddings=emb, documents=rec["documents"], metadatas=rec["metadatas"])
                    elif rec["op"] == "delete":
                        self.delete(ids=rec["ids"])
                    pos = good = nl + 1
                del raw
                if good < len(blob):               # drop the torn tail so that the next record starts on a fresh line
                    with open(jp, "r+b") as f:
                        f.truncate(good)
                if f32_size > f32_end:             # drop orphan vectors of an uncommitted record
                    with open(fp, "r+b") as f:
                        f.truncate(f32_end)
        finally:
            self._replaying = False
        sh = meta.get("xcd_shares")
        if sh and self._engine is not None and hasattr(self._engine, "xcd_shares"):
            try:
                self._engine.xcd_shares(sh)
            except Exception:      # (a header from another build: the shares are a hint, nothing more)
                pass


class PersistentClient:
    """chromadb.PersistentClient look-alike (reference app.py:58-59, create_chromadb_index.py:70-130,
    eval/run_eval.py:712-715): `PersistentClient(path).get_collection("rag_dpo_chunks")`.
    Collections are loaded into HBM on open; every add/update/delete is journalled when it returns, persist()/close()
    fold the journal into a snapshot."""

    def __init__(self, path: Optional[str] = None, device: int = 0, engine_factory=None, settings=None,
                 devices: Optional[Sequence[int]] = None, **_ignored):
        self.path = path
        self._device = device
        if engine_factory is None and devices is not None and len(devices) > 1:
            from .multi_device import multi_device_factory
            engine_factory = multi_device_factory(devices)
        self._factory = engine_factory
        self._cols: Dict[str, Collection] = {}
        if path and os.path.isdir(path):
            for name in sorted(os.listdir(path)):
                d = os.path.join(path, name)
                if os.path.exists(os.path.join(d, "collection.json")):
                    c = Collection(name, device=device, engine_factory=engine_factory, _client=self)
                    c._load(d)
                    c._dir = d
                    self._cols[name] = c

    def create_collection(self, name: str, metadata: Optional[dict] = None, get_or_create: bool = False, **_ignored):
        if name in self._cols:
            if get_or_create:
                return self._cols[name]
            raise ValueError(f"Collection {name} already exists")
        c = Collection(name, metadata=metadata, device=self._device, engine_factory=self._factory, _client=self)
        self._cols[name] = c
        if self.path:
            c._dir = os.path.join(self.path, name)
            c._write_header(c._dir, 0)
        return c

    def get_collection(self, name: str, **_ignored) -> Collection:
        if name not in self._cols:
            raise NotFoundError(f"Collection {name} does not exist.")
        return self._cols[name]

    def get_or_create_collection(self, name: str, metadata: Optional[dict] = None, **_ignored) -> Collection:
        return self.create_collection(name, metadata=metadata, get_or_create=True)

    def delete_collection(self, name: str):
        if name not in self._cols:
            raise NotFoundError(f"Collection {name} does not exist.")
        c = self._cols.pop(name)
        c._drop_masks()
        if c._engine is not None and hasattr(c._engine, "close"):
            c._engine.close()
        if self.path:
            d = os.path.join(self.path, name)
            if os.path.isdir(d):
                for fn in os.listdir(d):
                    if fn.startswith(("collection.json", "snap", "journal", "records.jsonl", "embeddings.f32.npy")):
                        os.remove(os.path.join(d, fn))
                try:
                    os.rmdir(d)
                except OSError:
                    pass

    def list_collections(self):
        return list(self._cols.values())

    def _rename(self, old: str, new: str):
        if new in self._cols:
            raise ValueError(f"Collection {new} already exists")
        c = self._cols[new] = self._cols.pop(old)
        if self.path and c._dir and os.path.isdir(c._dir):
            nd = os.path.join(self.path, new)
            os.rename(c._dir, nd)
            c._dir = nd

    def persist(self):
        if not self.path:
            return
        os.makedirs(self.path, exist_ok=True)
        for name, c in self._cols.items():
            c._dir = os.path.join(self.path, name)
            c._save(c._dir)

    def close(self):
        self.persist()

    def heartbeat(self) -> int:
        import time
        return int(time.time() * 1e9)


def Client(**kw):   # chromadb.Client(): in-memory
    return PersistentClient(path=None, **kw)


def import_collection(src, client: PersistentClient, name: Optional[str] = None, page: int = 5000) -> Collection:
    """One-shot import of an existing Chroma collection. `src` is anything Chroma-shaped, e.g.
    `chromadb.PersistentClient(path="data/vectordb/chromadb").get_collection("rag_dpo_chunks")` on a machine that has
    chromadb: it is paged through `get(limit=, offset=, include=[embeddings, documents, metadatas])` (the paging the
    reference itself uses, bm25_index.py:211-215) and re-added here, insertion order kept. chromadb is not imported by
    this package."""
    name = name or getattr(src, "name", "rag_dpo_chunks")
    meta = dict(getattr(src, "metadata", None) or {"hnsw:space": "cosine"})
    dst = client.create_collection(name=name, metadata=meta)
    total, off = src.count(), 0
    while off < total:
        g = src.get(limit=page, offset=off, include=["embeddings", "documents", "metadatas"])
        if not g["ids"]:
            break
        dst.add(ids=g["ids"], embeddings=np.asarray(g["embeddings"], dtype=np.float32), documents=g.get("documents"),
                metadatas=g.get("metadatas"))
        off += len(g["ids"])
    client.persist()
    return dst
