"""Latency of the drop-in boundary at the reference's own shape (config C1): Collection.query with Python lists in and
out (so PCIe and the Python result assembly are inside), 16,919 rows x 1024, n_results = 50, one query vector per call
(reference src/rag/retriever.py:215-220), then the batched form the retriever counterpart uses (4 reformulations, one call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rag_dpo_amd import synth
from rag_dpo_amd.collection import Collection

n, dim = 16919, 1024
emb = synth.make_corpus(n, dim)
col = Collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
nat = ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE"]
for a in range(0, n, 5000):
    b = min(n, a + 5000)
    col.add(ids=[f"chunk_{i}" for i in range(a, b)], embeddings=emb[a:b], documents=[f"doc {i} " * 40 for i in range(a, b)],
            metadatas=[{"document_id": f"d{i % 2000}", "document_path": f"p{i % 2000}.html", "document_nature": nat[i % 4],
                        "chunk_nature": nat[i % 4], "chunk_index": i % 9, "heading": f"h{i}", "page_info": "", "confidence": 0.5,
                        "method": "llm", "word_count": 120, "sectors": "", "file_type": "html", "title": f"t{i % 2000}",
                        "source": "CNIL", "source_type": "html", "is_priority": False, "source_url": f"https://cnil.fr/{i % 2000}",
                        "parent_url": ""} for i in range(a, b)])
q = synth.make_queries(8, dim, emb)
inc = ["documents", "metadatas", "distances"]
where = {"chunk_nature": {"$in": ["GUIDE", "DOCTRINE"]}}

def timeit(f, reps=200):
    for _ in range(10):
        f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t) / reps * 1e3

one = [q[0].tolist()]
four = [q[i].tolist() for i in range(4)]
print("query 1 x top-50, no where      : %.3f ms" % timeit(lambda: col.query(query_embeddings=one, n_results=50, include=inc)))
print("query 1 x top-50, where $in     : %.3f ms" % timeit(lambda: col.query(query_embeddings=one, n_results=50, where=where, include=inc)))
print("query 4 x top-50 in one call    : %.3f ms" % timeit(lambda: col.query(query_embeddings=four, n_results=50, include=inc)))
print("query 1 x top-50, ids+distances : %.3f ms" % timeit(lambda: col.query(query_embeddings=one, n_results=50, include=["distances"])))
qn = np.ascontiguousarray(q[:1])
print("engine.search 1 x top-50 (numpy): %.3f ms" % timeit(lambda: col._engine.search(qn, 50)))

if "--online" in sys.argv:
    # the reference's whole online step for one question: embed_query, then collection.query (src/rag/retriever.py:150-154, 215-220),
    # and this repo's retriever form: the question's <= 4 sub-queries embedded together and searched in one call
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    texts = synth.query_texts(16)
    for name, packed in (("module forward", False), ("librdx kernels + graph replay (default)", True)):
        p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=64)
        p.packed_forward = packed
        p.load()
        it = [0]
        def one_q():
            it[0] += 1
            v = p.embed([texts[it[0] % 8]])
            return col.query(query_embeddings=v, n_results=50, where=where, include=inc)
        def four_q():
            it[0] += 1
            a = (it[0] % 3) * 4
            v = p.embed(texts[a:a + 4])
            return col.query(query_embeddings=v, n_results=50, where=where, include=inc)
        for _ in range(30):
            one_q(); four_q()
        print("embed([question]) + query, where $in, top-50      [%s]: %.3f ms" % (name, timeit(one_q, 100)))
        print("embed(4 sub-queries) + query in one call, top-50  [%s]: %.3f ms" % (name, timeit(four_q, 100)))
