#!/usr/bin/env python3
"""Captures tests/golden/cosine_golden.{json,npy}: the ONE cosine the reference computes in its own code —
`TopicMatcher.similarity` (/root/reference/src/utils/rgpd_topics.py:167-177: `float(np.dot(vec_a, vec_b))` on the embedding
provider's unit vectors; the same expression at eval/run_eval.py:397-401) — run HERE, from the imported reference module (never
copied), on seeded d = 1024 unit vectors handed to it by a fake embedding provider. That makes the SCORE of this repo's oracle
(and of the HIP path) a reference-run fact. What stays Chroma's contract (chromadb==1.4.1, absent offline) and therefore
unpinned: `distance = 1 - cos`, the top-k order and the tie rule.

The reference does not travel to the GPU box: the committed fixture is data only — the vectors the fake provider returned
(float32 .npy, loaded with allow_pickle=False) and, per pair, the float the reference returned.

    python tests/golden/make_cosine_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from src.utils.rgpd_topics import TopicMatcher  # noqa: E402  (the reference)

DIM = 1024


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)     # what a provider with normalize_embeddings=True hands out: fp32, |v| = 1 +- ulp


def world():
    """name -> unit vector, and the pairs to compare"""
    rng = np.random.default_rng(20261004)
    vec, pairs = {}, []
    for i in range(12):
        vec[f"random_{i}"] = unit(rng.standard_normal(DIM))
    for i in range(0, 12, 2):
        pairs.append((f"random_{i}", f"random_{i + 1}", "random"))
    pairs += [("random_0", "random_5", "random"), ("random_3", "random_8", "random")]
    # the same direction twice (a chunk stored twice / a query equal to a stored row): cos = 1 up to the unit vector's own rounding
    vec["dup_a"] = vec["random_2"].copy()
    pairs += [("random_2", "dup_a", "duplicate"), ("random_7", "random_7", "duplicate")]
    # orthogonal: axis vectors, and a Gram-Schmidt partner of a random vector
    e3, e700 = np.zeros(DIM), np.zeros(DIM)
    e3[3], e700[700] = 1.0, 1.0
    vec["axis_3"], vec["axis_700"] = unit(e3), unit(e700)
    a = rng.standard_normal(DIM)
    b = rng.standard_normal(DIM)
    b -= a * (a @ b) / (a @ a)
    vec["gs_a"], vec["gs_b"] = unit(a), unit(b)
    pairs += [("axis_3", "axis_700", "orthogonal"), ("gs_a", "gs_b", "orthogonal"), ("axis_3", "random_1", "axis-vs-random")]
    # antipodal
    vec["anti_4"] = (-vec["random_4"]).astype(np.float32)
    pairs += [("random_4", "anti_4", "antipodal"), ("axis_700", "anti_axis", "antipodal")]
    vec["anti_axis"] = unit(-e700)
    # near-duplicates: a row plus a small perturbation (the planted queries of SURVEY.md §8d, and tighter ones)
    for j, eps in enumerate((0.3, 1e-2, 1e-3, 1e-4)):
        base = vec[f"random_{8 + j % 4}"].astype(np.float64)
        vec[f"near_{j}"] = unit(base + eps * rng.standard_normal(DIM) / np.sqrt(DIM))
        pairs.append((f"random_{8 + j % 4}", f"near_{j}", f"near-duplicate eps={eps}"))
    # badly scaled components: a few large, many tiny (fp16 scan copy underflow territory; the exact score must not care)
    s = rng.standard_normal(DIM) * 1e-4
    s[:4] = [3.0, -2.0, 1.5, 0.5]
    t = rng.standard_normal(DIM) * 1e-4
    t[:4] = [2.5, -1.0, 2.0, -0.7]
    vec["spiky_a"], vec["spiky_b"] = unit(s), unit(t)
    pairs += [("spiky_a", "spiky_b", "spiky"), ("spiky_a", "random_9", "spiky-vs-random")]
    return vec, pairs


class FakeProvider:
    """stands where EmbeddingProvider stands (embed(texts) -> List[List[float]], reference src/utils/embedding_provider.py:118-147)"""

    def __init__(self, vec):
        self.vec, self.calls = vec, 0

    def embed(self, texts):
        self.calls += 1
        return [self.vec[t].tolist() for t in texts]       # Python floats, as the reference's provider returns them


def main():
    vec, pairs = world()
    names = sorted(vec)
    tm = TopicMatcher(embedding_provider=FakeProvider(vec))
    out = []
    for a, b, kind in pairs:
        sim = tm.similarity(a, b)                          # <- the reference's arithmetic
        assert isinstance(sim, float)
        out.append({"a": a, "b": b, "kind": kind, "ia": names.index(a), "ib": names.index(b), "similarity": sim})
    # the order a top-k must reproduce: for a few query vectors, the reference's similarity to EVERY stored vector
    rankings = [{"q": qn, "iq": names.index(qn), "similarities": [tm.similarity(qn, n) for n in names]}
                for qn in ("random_0", "random_9", "axis_700", "random_11", "spiky_a")]
    mat = np.stack([vec[n] for n in names]).astype(np.float32)
    np.save(os.path.join(HERE, "cosine_golden.npy"), mat)
    with open(os.path.join(HERE, "cosine_golden.json"), "w") as f:
        json.dump({"_source": "TopicMatcher.similarity of /root/reference/src/utils/rgpd_topics.py:167-177, imported and run by "
                              "tests/golden/make_cosine_golden.py with a fake embedding provider returning the rows of cosine_golden.npy",
                   "dim": DIM, "names": names, "pairs": out, "rankings": rankings}, f, indent=1)
    print(f"{len(out)} pairs over {len(names)} vectors; similarities {min(p['similarity'] for p in out):+.6f} .. "
          f"{max(p['similarity'] for p in out):+.6f}")


if __name__ == "__main__":
    main()
