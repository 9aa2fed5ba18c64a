"""The vectorised where-filter (rag_dpo_amd/where.py: typed columns, dictionary-coded strings, numpy masks) against a
row-at-a-time restatement of the same rules (module docstring of where.py: comparisons inside one type only, a missing
key matches $ne/$nin and nothing else) on seeded random metadata and random filter trees."""
import random

import numpy as np

from rag_dpo_amd.where import Column, evaluate, kind_of

KEYS = ["chunk_nature", "source", "chunk_index", "confidence", "is_priority", "tag_rh", "absent_everywhere"]
POOL = {
    "chunk_nature": ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE", ""],
    "source": ["CNIL", "ENTREPRISE"],
    "chunk_index": [0, 1, 2, 3, 7, -1, "OPERATIONNEL"],        # mixed kinds in one column, as the reference's default produces
    "confidence": [0.5, 0.91, 0.73, 1.0, 1],                    # float and int 1 are different values
    "is_priority": [True, False],
    "tag_rh": [True],
}


def row_matches(meta: dict, w: dict) -> bool:
    (key, val), = w.items()
    if key == "$and":
        return all(row_matches(meta, x) for x in val)
    if key == "$or":
        return any(row_matches(meta, x) for x in val)
    op, operand = next(iter(val.items())) if isinstance(val, dict) else ("$eq", val)
    have = key in meta
    v = meta.get(key)
    same = lambda a, b: kind_of(a) == kind_of(b) and a == b
    if op == "$eq":
        return have and same(v, operand)
    if op == "$ne":
        return not (have and same(v, operand))
    if op == "$in":
        return have and any(same(v, x) for x in operand)
    if op == "$nin":
        return not (have and any(same(v, x) for x in operand))
    if not have or kind_of(v) != kind_of(operand):
        return False
    return {"$gt": v > operand, "$gte": v >= operand, "$lt": v < operand, "$lte": v <= operand}[op]


def random_where(rng: random.Random, depth: int = 0) -> dict:
    if depth < 3 and rng.random() < 0.35:
        return {rng.choice(["$and", "$or"]): [random_where(rng, depth + 1) for _ in range(rng.randint(2, 3))]}
    key = rng.choice(KEYS)
    pool = POOL.get(key, ["x", 1])
    op = rng.choice(["plain", "$eq", "$ne", "$in", "$nin", "$gt", "$gte", "$lt", "$lte"])
    if op == "plain":
        return {key: rng.choice(pool)}
    if op in ("$in", "$nin"):
        first = rng.choice(pool)
        same_kind = [x for x in pool if kind_of(x) == kind_of(first)]
        return {key: {op: rng.sample(same_kind, rng.randint(1, len(same_kind)))}}
    if op in ("$gt", "$gte", "$lt", "$lte"):
        nums = [x for x in pool if kind_of(x) in (2, 3)] or [1]
        return {key: {op: rng.choice(nums)}}
    return {key: {op: rng.choice(pool)}}


def test_vectorised_filter_equals_row_at_a_time_rules():
    rng = random.Random(20260401)
    n = 700
    metas = []
    for _ in range(n):
        m = {}
        for key, pool in POOL.items():
            if rng.random() < (0.15 if key == "tag_rh" else 0.8):
                m[key] = rng.choice(pool)
        metas.append(m)
    cols = {}
    for r, m in enumerate(metas):
        for k, v in m.items():
            cols.setdefault(k, Column(n)).set(r, v)
    for _ in range(400):
        w = random_where(rng)
        got = evaluate(w, cols, n)
        want = np.array([row_matches(m, w) for m in metas])
        assert (got == want).all(), (w, np.flatnonzero(got != want)[:5])
