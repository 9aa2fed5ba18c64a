"""Developer: where an ingest batch spends its wall time — host tokenising, host preparing / enqueueing the forward, the wait for the GPU
inside collection.add (it synchronises the device), and the add's own host work. Same workload as tools/ingest_bench.py, n chunks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_dpo_amd import synth
from rag_dpo_amd.collection import Client, Collection
from rag_dpo_amd.embedding_provider import EmbeddingProvider
from rag_dpo_amd.indexer import ChromaDBIndexer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(2026)
lens = np.clip(np.exp(rng.normal(5.5, 0.65, size=n)), 64, 1024).astype(np.int64)
words = synth._WORDS
chunks = [{"chunk_id": f"chunk_{i}", "document_id": f"doc{i // 9}", "document_path": f"data/raw/cnil/html/p{i // 9}.html", "heading": f"Section {i % 50} RGPD",
           "text": " ".join(words[int(j)] for j in rng.integers(0, len(words), int(lens[i]) - 5)), "chunk_nature": "GUIDE", "chunk_index": i % 9, "confidence": 0.9}
          for i in range(n)]
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=int(os.environ.get("RDX_ENC_BATCH", "64"))).load()
T = {}
def timed(obj, name, label, sync=False):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        if sync: torch.cuda.synchronize()
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
timed(p, "_tokenizer", "tokenise (host)")
timed(p._packed, "cls", "forward: host prep + enqueue (async)")
ix = ChromaDBIndexer(Client(), p, device_embeddings=True)
ix.init_chromadb("reset")
ix.index_chunks(chunks[:300], 100)
torch.cuda.synchronize()
ix.init_chromadb("reset")
T.clear()
timed(ix.collection, "add", "collection.add (device sync + K1 + host records)")
timed(ix, "generate_embeddings", "embed_device as a whole")
t0 = time.perf_counter()
ix.index_chunks(chunks, 100)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"{n} chunks in {tot:.3f} s = {n / tot:.0f} chunks/s, {n // 100} batches")
for k, v in T.items():
    print(f"  {k:55s} {v * 1e3 / (n / 100):8.2f} ms per batch of 100")
print(f"  {'everything':55s} {tot * 1e3 / (n / 100):8.2f} ms per batch of 100")
