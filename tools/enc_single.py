"""Developer: latency of encoding ONE question (the reference's online path: embed_query, then collection.query) — module forward,
packed forward with librdx's kernels, and the same replayed as a HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
texts = synth.query_texts(64)
prov = {}
for name, packed, graphs in (("module", False, False), ("fused", True, False), ("graph", True, None)):
    p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=64)
    p.packed_forward, p.encoder_graphs = packed, graphs
    prov[name] = p.load()
sample = texts[:8]
for name, p in prov.items():
    for t in sample:                     # warm-up (and capture: a shape is captured the second time it is seen)
        for _ in range(3): p.embed_device([t])
    torch.cuda.synchronize()
    lat = []
    for rep in range(5):
        for t in sample:
            t0 = time.perf_counter(); v = p.embed_device([t]); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    t0 = time.perf_counter()
    for rep in range(5):
        for t in sample: v = p.embed([t])
    emb = (time.perf_counter() - t0) / 40 * 1e3
    four = [sample[i:i + 4] for i in (0, 4)]
    for b4 in four:
        for _ in range(3): p.embed_device(b4)
    torch.cuda.synchronize()
    l4 = []
    for rep in range(10):
        for b4 in four:
            t0 = time.perf_counter(); v = p.embed_device(b4); torch.cuda.synchronize(); l4.append((time.perf_counter() - t0) * 1e3)
    l4.sort()
    print(f"{name:>7}: embed_device([four questions]) + synchronise: median {l4[len(l4)//2]:.3f} ms, min {l4[0]:.3f}")
    print(f"{name:>7}: embed_device([one question]) + synchronise: median {lat[len(lat)//2]:.3f} ms, min {lat[0]:.3f}, p90 {lat[int(len(lat)*0.9)]:.3f}; embed([q]) (normalise + list) {emb:.3f} ms", flush=True)
