"""TEST-ONLY engine: the CPU oracle behind the engine interface of rag_dpo_amd.collection / sharded.
Lets the host logic (ids, metadata, where filters, paging, sharded merge orchestration) run without a GPU.
The product never imports this file; its only engine is rag_dpo_amd.engine.HipIndex."""
import numpy as np

from oracle import oracle as O


class OracleEngine:
    def __init__(self, dim, device=0):
        self.dim = dim
        self.device = device
        self.rows = np.zeros((0, dim), dtype=np.float32)

    def __len__(self):
        return self.rows.shape[0]

    def _check(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError("bad shape")
        if not np.isfinite(x).all():
            raise ValueError("embeddings contain NaN or Inf")
        return x

    def add(self, x):
        self.rows = np.concatenate([self.rows, O.normalize_rows(self._check(x))])

    def add_stored(self, x):
        self.rows = np.concatenate([self.rows, self._check(x)])

    def update(self, ids, x):
        self.rows[np.asarray(ids, dtype=np.int64)] = O.normalize_rows(self._check(x))

    def get(self, ids):
        return self.rows[np.asarray(ids, dtype=np.int64)].copy()

    def compact(self, keep):
        self.rows = self.rows[np.asarray(keep, dtype=np.int64)].copy()

    def search(self, q, k, allow_bits=None):
        q = self._check(q)
        allow = None
        if allow_bits is not None:
            n = len(self)
            allow = np.unpackbits(np.asarray(allow_bits, dtype=np.uint32).view(np.uint8), bitorder="little")[:n].astype(bool)
        if len(self) == 0:
            B = q.shape[0]
            return (np.full((B, k), -np.inf, np.float32), np.full((B, k), -1, np.int64), np.zeros(B, np.int32))
        return O.cosine_topk(self.rows, q, k, allow)

    def close(self):
        pass


def factory(dim, device=0):
    return OracleEngine(dim, device)
