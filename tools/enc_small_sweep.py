"""Developer: encode latency of 1 / 2 / 4 / 8 questions (graph replay) by the token count up to which the projections take librdx's
weight-streaming kernel (_PackedEncoder.SMALL_TOKENS; 0 = always the BLAS library)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider, _PackedEncoder
texts = synth.query_texts(64)
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=64).load()
for small in (0, 32, 64, 128, 256):
    p._packed.SMALL_TOKENS = small
    p._packed._graph.clear(); p._packed._seen.clear()
    line = [f"SMALL_TOKENS {small:3d}:"]
    for nb in (1, 2, 4, 8):
        batches = [texts[i:i + nb] for i in range(0, 32, nb)][:6]
        for b in batches:
            for _ in range(3): p.embed_device(b)
        torch.cuda.synchronize()
        lat = []
        for rep in range(6):
            for b in batches:
                t0 = time.perf_counter(); p.embed_device(b); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        line.append(f"{nb} q {lat[len(lat)//2]:.3f} ms ({p.last_encode_stats['tokens_real']} tok)")
    print("  ".join(line), flush=True)
