"""The small deterministic world shared by make_retriever_golden.py (capture, runs the IMPORTED reference retriever
in the build container) and tests/test_retriever_golden.py (replay, runs only this repo's code)."""
import zlib

import numpy as np

from rag_dpo_amd import synth
from rag_dpo_amd.collection import Collection

N, DIM = 400, 64
NAT = ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE"]


def corpus():
    return synth.make_corpus(N, DIM)


def build_collection(engine_factory):
    emb = corpus()
    col = Collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"}, engine_factory=engine_factory)
    metas = []
    for i in range(N):
        m = {"document_path": f"cnil/doc_{i % 37}.html", "chunk_nature": NAT[i % 4], "chunk_index": i % 9,
             "confidence": "high" if i % 3 else "medium", "source": "ENTREPRISE" if i % 11 == 0 else "CNIL"}
        if i % 5 == 0:   # several documents share a URL up to scheme/www -> exercises the URL de-duplication
            m["source_url"] = ("https://www." if i % 2 else "http://") + f"cnil.fr/page_{(i % 37) % 6}"
        if i % 11 == 0 and i % 2 == 0:
            m["tag_rh"] = True
        metas.append(m)
    col.add(ids=[f"chunk_{i:04d}" for i in range(N)], documents=[f"texte du chunk {i}" for i in range(N)],
            embeddings=emb.tolist(), metadatas=metas)
    return col


class HashEmbedder:
    """stands in for EmbeddingProvider: deterministic text -> vector near some corpus row"""
    model_name = "hash-embedder"

    def __init__(self, poison_exact=(), poison_sub=()):
        self.c = corpus()
        self.calls = []
        self.poison_exact, self.poison_sub = set(poison_exact), tuple(poison_sub)

    def embed(self, texts):
        self.calls.append(list(texts))
        out = []
        for t in texts:
            if t in self.poison_exact or any(p in t for p in self.poison_sub):
                out.append([float("nan")] * DIM)      # collection.query raises on it -> that sub-query is skipped
                continue
            h = zlib.crc32(t.encode("utf-8"))
            v = self.c[h % N] + 0.8 * np.random.default_rng(h).standard_normal(DIM).astype(np.float32)
            out.append((v / np.linalg.norm(v)).astype(np.float32).tolist())
        return out


def expander(question):
    """stands in for QueryExpander.expand (an LLM call in the reference): original first, then 3 reformulations"""
    return [question, question + " (reformulation juridique)", "obligations " + question, question.lower() + " sanction"]


CASES = [
    {"query": "Quelle est la durée de conservation des données de vidéosurveillance ?", "where": None, "n_candidates": 40},
    {"query": "registre des traitements sous-traitant", "where": {"chunk_nature": {"$in": ["GUIDE", "DOCTRINE"]}}, "n_candidates": 60},
    {"query": "transfert hors union européenne",
     "where": {"$and": [{"chunk_nature": {"$in": ["GUIDE", "SANCTION"]}},
                        {"$or": [{"source": {"$ne": "ENTREPRISE"}}, {"tag_rh": True}]}]}, "n_candidates": 25},
]

# sub-queries whose collection.query raises (a NaN embedding): the reference logs and skips them
# (src/rag/retriever.py:221-223, 386-388); always with the expander (4 sub-queries)
FAIL_CASES = [
    {"query": CASES[0]["query"], "where": None, "n_candidates": 40, "poison_exact": [], "poison_sub": ["obligations "]},
    {"query": CASES[1]["query"], "where": CASES[1]["where"], "n_candidates": 60, "poison_exact": [CASES[1]["query"]], "poison_sub": []},
    {"query": CASES[2]["query"], "where": CASES[2]["where"], "n_candidates": 25, "poison_exact": [],
     "poison_sub": ["(reformulation", "obligations ", " sanction"]},
    {"query": CASES[0]["query"], "where": None, "n_candidates": 40, "poison_exact": [], "poison_sub": ["onn"]},   # every sub-query fails
]
