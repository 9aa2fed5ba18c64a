"""Ingest throughput of the write side of the hot path: the reference's `ChromaDBIndexer.index_chunks` loop
(src/processing/create_chromadb_index.py:300-387: batches of 100 chunks -> `heading\\n\\ntext` -> embed -> collection.add) through this
repo's mirror (rag_dpo_amd/indexer.py) with `device_embeddings=True`: encoder output -> K1 -> HBM, never Python floats.

16 919 synthetic chunks (the reference's corpus size, README.en.md:300-305) whose texts have a stated token-length distribution:
log-normal, clipped to [64, 1024] tokens, mean ~300 (the reference's chunks are up to ~5 K characters, embedding_provider.py:30-31).
XLM-R-large architecture, random-init fp16, hashing tokenizer (no BGE-M3 weights offline: value parity unpinned; shape and cost faithful).

Prints ONE JSON line: chunks/s, tokens/s, forward TFLOP/s (2 * 303 M * real tokens, the figure SURVEY.md §8a3 uses) against the 2.5 PF
MFMA peak, tokens_real / tokens_padded. The GPU-time split by kernel family comes from running this under rocprofv3 --kernel-trace
(tools/r04_ingest.sh classifies the kernel names).   python tools/ingest_bench.py [n_chunks] [--attn torch|mfma]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_dpo_amd import synth
from rag_dpo_amd.collection import Client
from rag_dpo_amd.embedding_provider import EmbeddingProvider
from rag_dpo_amd.indexer import ChromaDBIndexer

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16919
rng = np.random.default_rng(2026)
lens = np.clip(np.exp(rng.normal(5.5, 0.65, size=n)), 64, 1024).astype(np.int64)     # tokens per chunk incl. <s> </s>
words = synth._WORDS
NAT = ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE"]
chunks = []
for i in range(n):
    m = int(lens[i]) - 2 - 3                      # the heading contributes 3 words
    body = " ".join(words[int(j)] for j in rng.integers(0, len(words), m))
    chunks.append({"chunk_id": f"chunk_{i}", "document_id": f"doc{i // 9}", "document_path": f"data/raw/cnil/html/p{i // 9}.html",
                   "heading": f"Section {i % 50} RGPD", "text": body, "chunk_nature": NAT[i % 4], "chunk_index": i % 9, "confidence": 0.9})
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=int(os.environ.get("RDX_ENC_BATCH", "64")))
p.load()
ix = ChromaDBIndexer(Client(), p, device_embeddings=True)
ix.init_chromadb("reset")
ix.index_chunks(chunks[:300], batch_size=100)     # warm-up (library workspaces, clocks)
torch.cuda.synchronize()
ix.init_chromadb("reset")
ix.stats.update(chunks_indexed=0, errors=0)
tok_real = tok_pad = 0
orig = p._encode_raw
def counted(texts):
    global tok_real, tok_pad
    out = orig(texts)
    tok_real += p.last_encode_stats["tokens_real"]; tok_pad += p.last_encode_stats["tokens_padded"]
    return out
p._encode_raw = counted
t0 = time.perf_counter()
ix.index_chunks(chunks, batch_size=100)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert ix.stats["chunks_indexed"] == n and ix.collection.count() == n, ix.stats
flops = 2.0 * 303e6 * tok_real
print(json.dumps({"workload": f"ingest: {n} synthetic chunks, tokens log-normal clipped to [64, 1024] (mean {lens.mean():.0f}, median {int(np.median(lens))}, max {int(lens.max())}), "
                              "batches of 100 through ChromaDBIndexer(device_embeddings=True), provider batch_size %d" % p.batch_size,
                  "chunks_per_s": round(n / dt, 1), "tokens_per_s": round(tok_real / dt, 1), "seconds": round(dt, 3),
                  "forward_tflops": round(flops / dt / 1e12, 1), "frac_of_2.5PF": round(flops / dt / 1e12 / 2500.0, 4),
                  "tokens_real": int(tok_real), "tokens_padded": int(tok_pad), "tokens_real_over_padded": round(tok_real / max(1, tok_pad), 4),
                  "attention": os.environ.get("RDX_ENC_LONG_ATTN", "default"),
                  "model": "XLM-R-large (BGE-M3 architecture), random-init fp16, hashing tokenizer; encoder value parity unpinned"}), flush=True)
