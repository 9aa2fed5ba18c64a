"""Developer: where the host time of Collection.query goes at the reference's shape (one query vector as a Python list, n_results=50)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rag_dpo_amd import synth
from rag_dpo_amd.collection import Collection
n, dim = 16919, 1024
emb = synth.make_corpus(n, dim)
col = Collection("c", metadata={"hnsw:space": "cosine"})
for a in range(0, n, 5000):
    b = min(n, a + 5000)
    col.add(ids=[f"chunk_{i}" for i in range(a, b)], embeddings=emb[a:b], documents=[f"doc {i} " * 40 for i in range(a, b)],
            metadatas=[{"document_id": f"d{i % 2000}", "chunk_nature": "GUIDE", "chunk_index": i % 9, "heading": f"h{i}", "confidence": 0.5,
                        "word_count": 120, "is_priority": False, "source_url": f"https://cnil.fr/{i % 2000}"} for i in range(a, b)])
one = [synth.make_queries(1, dim, emb)[0].tolist()]
inc = ["documents", "metadatas", "distances"]
f = lambda: col.query(query_embeddings=one, n_results=50, include=inc)
for _ in range(50): f()
t = time.perf_counter()
for _ in range(2000): f()
print("per query %.1f us" % ((time.perf_counter() - t) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
