"""Chroma `where` pre-filter evaluated on host columnar metadata -> row bitmap for the device scan.

Grammar actually used by the reference: {field: scalar} (eq), {field: {"$in": [...]}}
(pages/1_💬_Chat.py:247), {field: {"$ne": v}}, {"tag_x": True}, {"$or": [...]}, {"$and": [...]}
(src/rag/pipeline.py:35-71). The remaining Chroma comparison operators ($gt/$gte/$lt/$lte/$nin/$eq) are
accepted too. Semantics = filter-then-knn (reference tasks/lessons.md:53-57).

Chroma stores metadata values typed (str / int / float / bool) and compares inside one type only: an int 1
does not match a bool True or a float 1.0. A missing key never matches $eq/$in/$gt/...; it does match
$ne/$nin. Values are str/int/float/bool only (reference src/processing/create_chromadb_index.py:339-360).
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional

import numpy as np

K_MISSING, K_STR, K_INT, K_FLOAT, K_BOOL = 0, 1, 2, 3, 4
_LOGICAL = ("$and", "$or")
_COMPARE = ("$eq", "$ne", "$gt", "$gte", "$lt", "$lte", "$in", "$nin")


def kind_of(v: Any) -> int:
    if isinstance(v, bool) or isinstance(v, np.bool_):
        return K_BOOL
    if isinstance(v, (int, np.integer)):
        return K_INT
    if isinstance(v, (float, np.floating)):
        return K_FLOAT
    if isinstance(v, str):
        return K_STR
    raise ValueError(f"metadata values must be str, int, float or bool, got {type(v).__name__}: {v!r}")


def check_meta(meta: Optional[dict]):
    """everything Column.set could object to, checked BEFORE anything is stored (so that a bad value in the middle of a
    batch cannot leave the host rows and the device rows of a collection out of step)"""
    if not meta:
        return
    for k, v in meta.items():
        if not isinstance(k, str):
            raise ValueError(f"Expected metadata key to be a str, got {k!r}")
        if v is None:
            continue
        kd = kind_of(v)
        if kd == K_INT and abs(int(v)) >= 2 ** 53:
            raise ValueError("integer metadata beyond 2^53 is not supported")
        if kd != K_STR:
            float(v)


class Column:
    """One metadata key over all rows: a kind tag per row + typed storage (strings dictionary-coded)."""

    def __init__(self, n: int = 0):
        self.kind = np.zeros(n, dtype=np.int8)
        self.num = np.zeros(n, dtype=np.float64)     # int / float / bool payload (ints exact below 2^53)
        self.code = np.full(n, -1, dtype=np.int32)   # string payload: index into vocab
        self.vocab: List[str] = []
        self._lookup: Dict[str, int] = {}

    def resize(self, n: int):
        old = self.kind.shape[0]
        if n <= old:
            self.kind, self.num, self.code = self.kind[:n], self.num[:n], self.code[:n]
            return
        cap = max(n, old + old // 2 + 16)
        for name, fill in (("kind", 0), ("num", 0.0), ("code", -1)):
            a = getattr(self, name)
            b = np.full(cap, fill, dtype=a.dtype)
            b[:old] = a
            setattr(self, name, b)

    def set(self, row: int, v: Any):
        if row >= self.kind.shape[0]:
            self.resize(row + 1)
        if v is None:
            self.kind[row] = K_MISSING
            return
        k = kind_of(v)
        self.kind[row] = k
        if k == K_STR:
            c = self._lookup.get(v)
            if c is None:
                c = len(self.vocab)
                self.vocab.append(v)
                self._lookup[v] = c
            self.code[row] = c
        else:
            if k == K_INT and abs(int(v)) >= 2 ** 53:
                raise ValueError("integer metadata beyond 2^53 is not supported")
            self.num[row] = float(v)

    def get(self, row: int):
        k = self.kind[row]
        if k == K_MISSING:
            return None
        if k == K_STR:
            return self.vocab[self.code[row]]
        if k == K_INT:
            return int(self.num[row])
        if k == K_FLOAT:
            return float(self.num[row])
        return bool(self.num[row])

    def take(self, rows: np.ndarray) -> "Column":
        c = Column(0)
        c.kind, c.num, c.code = self.kind[rows].copy(), self.num[rows].copy(), self.code[rows].copy()
        c.vocab, c._lookup = list(self.vocab), dict(self._lookup)
        return c

    # ---- predicates over rows [0, n) ---------------------------------------------------------
    def _eq(self, v: Any, n: int) -> np.ndarray:
        k = kind_of(v)
        same = self.kind[:n] == k
        if k == K_STR:
            c = self._lookup.get(v, -2)
            return same & (self.code[:n] == c)
        return same & (self.num[:n] == float(v))

    def _cmp(self, op: str, v: Any, n: int) -> np.ndarray:
        k = kind_of(v)
        if k not in (K_INT, K_FLOAT):
            raise ValueError(f"Expected operand value to be an int or a float for operator {op}, got {v!r}")
        same = self.kind[:n] == k
        x = self.num[:n]
        f = float(v)
        if op == "$gt":
            return same & (x > f)
        if op == "$gte":
            return same & (x >= f)
        if op == "$lt":
            return same & (x < f)
        return same & (x <= f)

    def test(self, op: str, v: Any, n: int) -> np.ndarray:
        if op == "$eq":
            return self._eq(v, n)
        if op == "$ne":
            return ~self._eq(v, n)
        if op in ("$in", "$nin"):
            if not isinstance(v, (list, tuple)) or len(v) == 0:
                raise ValueError(f"Expected where operand value to be a non-empty list for {op}, got {v!r}")
            kinds = {kind_of(x) for x in v}
            if len(kinds) != 1:
                raise ValueError(f"Expected where operand value to be a list of one type for {op}, got {v!r}")
            m = np.zeros(n, dtype=bool)
            for x in v:
                m |= self._eq(x, n)
            return m if op == "$in" else ~m
        return self._cmp(op, v, n)


def validate_where(where: Any):
    """Raises ValueError the way chromadb.api.types.validate_where does for malformed filters."""
    if not isinstance(where, dict):
        raise ValueError(f"Expected where to be a dict, got {where!r}")
    if len(where) != 1:
        raise ValueError(f"Expected where to have exactly one operator, got {where!r}")
    (key, val), = where.items()
    if not isinstance(key, str):
        raise ValueError(f"Expected where key to be a str, got {key!r}")
    if key.startswith("$") and key not in _LOGICAL:
        raise ValueError(f"Expected where key to be a metadata field or one of {_LOGICAL}, got {key}")
    if key in _LOGICAL:
        if not isinstance(val, list) or len(val) < 2:
            raise ValueError(f"Expected where value for {key} to be a list with at least two where expressions, got {val!r}")
        for w in val:
            validate_where(w)
        return
    if isinstance(val, dict):
        if len(val) != 1:
            raise ValueError(f"Expected operator expression to have exactly one operator, got {val!r}")
        (op, operand), = val.items()
        if op not in _COMPARE:
            raise ValueError(f"Expected where operator to be one of {_COMPARE}, got {op}")
        if op in ("$in", "$nin"):
            if not isinstance(operand, (list, tuple)) or len(operand) == 0:
                raise ValueError(f"Expected where operand value to be a non-empty list for {op}, got {operand!r}")
            if len({kind_of(x) for x in operand}) != 1:
                raise ValueError(f"Expected where operand value to be a list of one type for {op}, got {operand!r}")
        elif op in ("$gt", "$gte", "$lt", "$lte"):
            if kind_of(operand) not in (K_INT, K_FLOAT):
                raise ValueError(f"Expected operand value to be an int or a float for operator {op}, got {operand!r}")
        else:
            kind_of(operand)
    else:
        kind_of(val)


def evaluate(where: Optional[dict], columns: Dict[str, Column], n: int) -> Optional[np.ndarray]:
    """-> bool[n] (True = row passes) or None for "no filter"."""
    if where is None or where == {}:
        return None
    validate_where(where)
    return _eval(where, columns, n)


def _eval(where: dict, columns: Dict[str, Column], n: int) -> np.ndarray:
    (key, val), = where.items()
    if key == "$and":
        m = _eval(val[0], columns, n)
        for w in val[1:]:
            m = m & _eval(w, columns, n)
        return m
    if key == "$or":
        m = _eval(val[0], columns, n)
        for w in val[1:]:
            m = m | _eval(w, columns, n)
        return m
    if isinstance(val, dict):
        (op, operand), = val.items()
    else:
        op, operand = "$eq", val
    col = columns.get(key)
    if col is None:   # key absent everywhere: positive operators match nothing, $ne/$nin match everything
        return np.full(n, op in ("$ne", "$nin"), dtype=bool)
    return col.test(op, operand, n)


def pack_bits(mask: np.ndarray) -> np.ndarray:
    """bool[n] -> uint32 words in the layout include/rdx.h states (bit r&31 of word r>>5)."""
    n = mask.shape[0]
    padded = np.zeros(((n + 31) // 32) * 32, dtype=np.uint8)
    padded[:n] = mask
    return np.packbits(padded.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy()
