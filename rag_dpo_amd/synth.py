"""Deterministic synthetic corpora/queries of SURVEY.md §8(d) (tests, smoke and bench share them).

numpy path (small sizes, bit-reproducible across hosts): corpus chunk j of 65 536 rows comes from
default_rng([1234, j]); queries from default_rng(4321); 10 % of the queries are planted near a corpus row
(q = c_i + 0.3 * eps) and 1 % of the corpus rows are exact duplicates of other rows (tie rule).
torch path (bench sizes, generated in HBM): same recipe with torch.Generator seeds per chunk.
Rows are NOT normalised here — the index normalises on ingest, exactly as `collection.add` receives them.
"""
from __future__ import annotations

import numpy as np

CHUNK = 65536


def corpus_chunk(j: int, rows: int, dim: int) -> np.ndarray:
    return np.random.default_rng([1234, j]).standard_normal((rows, dim), dtype=np.float32)


def make_corpus(n: int, dim: int = 1024, duplicates: bool = True) -> np.ndarray:
    out = np.empty((n, dim), dtype=np.float32)
    for j, r0 in enumerate(range(0, n, CHUNK)):
        m = min(CHUNK, n - r0)
        out[r0:r0 + m] = corpus_chunk(j, m, dim)
    if duplicates and n >= 200:
        rng = np.random.default_rng(99)
        n_dup = n // 100
        dst = rng.choice(n, size=n_dup, replace=False)
        src = rng.integers(0, n, size=n_dup)
        out[dst] = out[src]
    return out


def make_queries(b: int, dim: int = 1024, corpus: np.ndarray | None = None) -> np.ndarray:
    rng = np.random.default_rng(4321)
    q = rng.standard_normal((b, dim), dtype=np.float32)
    if corpus is not None and corpus.shape[0] > 0:
        n_plant = max(1, b // 10)
        which = rng.choice(b, size=n_plant, replace=False)
        rows = rng.integers(0, corpus.shape[0], size=n_plant)
        eps = rng.standard_normal((n_plant, dim), dtype=np.float32)
        c = corpus[rows]
        c = c / np.linalg.norm(c, axis=1, keepdims=True)
        q[which] = c + 0.3 * eps / np.sqrt(dim)
    return q


def torch_corpus_chunk(j: int, rows: int, dim: int, device, duplicates: bool = True):
    """chunk j of the bench-size corpus, generated in HBM: `rows` N(0,1) rows from seed (1234, j); 1 % of them are then
    overwritten with exact copies of other rows OF THE SAME CHUNK (the tie rule at bench size), so any
    shard can still be regenerated chunk by chunk without its neighbours"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(1234 * 1000003 + j)
    out = torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)
    if duplicates and rows >= 200:
        n_dup = rows // 100
        dst = torch.randperm(rows, generator=g, device=device)[:n_dup]
        src = torch.randint(0, rows, (n_dup,), generator=g, device=device)
        out[dst] = out[src]          # advanced indexing reads the pre-assignment values
    return out


def torch_corpus_row(i: int, total_rows: int, dim: int, device):
    """raw row i of the bench corpus as build_shard stores it (whole chunk regenerated: ~1 ms on the GPU)"""
    j = i // CHUNK
    return torch_corpus_chunk(j, min(CHUNK, total_rows - j * CHUNK), dim, device)[i % CHUNK].clone()


def torch_queries(b: int, dim: int, device, total_rows: int = 0, return_planted: bool = False):
    """b N(0,1) queries from seed 4321; with total_rows > 0, 10 % of them are planted next to a corpus row:
    q = c_i/|c_i| + 0.3 * eps/sqrt(dim) — identical on every rank (each regenerates the rows it needs)"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    q = torch.randn((b, dim), generator=g, device=device, dtype=torch.float32)
    planted = []
    if total_rows > 0 and b >= 10:
        n_plant = b // 10
        which = torch.randperm(b, generator=g, device=device)[:n_plant].tolist()
        rows = torch.randint(0, total_rows, (n_plant,), generator=g, device=device).tolist()
        eps = torch.randn((n_plant, dim), generator=g, device=device, dtype=torch.float32)
        cache = {}
        for t, (qi, ri) in enumerate(zip(which, rows)):
            j = ri // CHUNK
            if j not in cache:
                cache.clear()        # one chunk (268 MB at d = 1024) at a time
                cache[j] = torch_corpus_chunk(j, min(CHUNK, total_rows - j * CHUNK), dim, device)
            c = cache[j][ri % CHUNK]
            q[qi] = c / c.norm() + 0.3 * eps[t] / (dim ** 0.5)
            planted.append((qi, ri))
        cache.clear()
    return (q, planted) if return_planted else q


# ---- an embedding-like corpus ---------------------------------------------------------------------------------
# Real sentence embeddings are not i.i.d. noise: they share a mean direction (anisotropy: two unrelated BGE-M3 vectors have a
# cosine around 0.4 - 0.6), and the reference stores a document's chunks back to back (`heading\n\ntext` of consecutive chunks,
# reference src/processing/create_chromadb_index.py:300-387), i.e. runs of near-duplicates. Here: row = mu + d_doc + sigma_doc * eps
# with |mu| = |d_doc| = 1, eps ~ N(0, I / dim) (so unrelated rows have cosine ~ 0.5), EMBED_DOCS documents of total_rows / EMBED_DOCS
# CONTIGUOUS chunks each, sigma_doc uniform in [0.2, 0.6] (chunks of one document: cosine 0.85 - 0.98); a query is a document's
# vector seen through noise 0.3. Deterministic per chunk of 65 536 rows like the N(0,1) corpus.
EMBED_DOCS = 2000


def _embed_tables(dim: int, device):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(777)
    mu = torch.randn((dim,), generator=g, device=device, dtype=torch.float32)
    mu = mu / mu.norm()
    docs = torch.randn((EMBED_DOCS, dim), generator=g, device=device, dtype=torch.float32)
    docs = docs / docs.norm(dim=1, keepdim=True)
    sigma = 0.2 + 0.4 * torch.rand((EMBED_DOCS,), generator=g, device=device, dtype=torch.float32)
    return mu, docs, sigma


def torch_embedlike_chunk(j: int, rows: int, dim: int, device, total_rows: int):
    import torch
    mu, docs, sigma = _embed_tables(dim, device)
    doc_len = max(1, -(-total_rows // EMBED_DOCS))
    r0 = j * CHUNK
    doc = torch.clamp((torch.arange(r0, r0 + rows, device=device) // doc_len), max=EMBED_DOCS - 1)
    g = torch.Generator(device=device)
    g.manual_seed(5678 * 1000003 + j)
    eps = torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32) * (dim ** -0.5)
    return mu[None, :] + docs[doc] + sigma[doc][:, None] * eps


def torch_embedlike_queries(b: int, dim: int, device):
    """b queries, each near one document: mu + d_doc + 0.3 * eps; -> (queries, document of each)"""
    import torch
    mu, docs, _ = _embed_tables(dim, device)
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    which = torch.randint(0, EMBED_DOCS, (b,), generator=g, device=device)
    eps = torch.randn((b, dim), generator=g, device=device, dtype=torch.float32) * (dim ** -0.5)
    return mu[None, :] + docs[which] + 0.3 * eps, which


_WORDS = ("durée conservation données personnelles traitement registre sous-traitant responsable AIPD analyse impact consentement "
          "cookies traceurs vidéosurveillance salariés transfert hors union européenne violation notification CNIL délégué "
          "protection base légale intérêt légitime droit accès effacement portabilité sanction mise en demeure sécurité").split()


def query_texts(b: int, seed: int = 4321) -> list:
    """b short question-like strings (8-24 words) for the encode leg of config C5; deterministic"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(b):
        n = int(rng.integers(8, 25))
        out.append("Quelle " + " ".join(_WORDS[int(i)] for i in rng.integers(0, len(_WORDS), n)) + " ?")
    return out
