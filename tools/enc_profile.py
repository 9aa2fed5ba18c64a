"""Developer: where the c5 query-encode time goes (tokenise on the host, forward on the GPU), 1024 short texts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=b).load()
texts = synth.query_texts(1024)
for _ in range(3):
    p.embed_device(texts)
torch.cuda.synchronize()
def timed(f, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
ms_tok, enc = timed(lambda: p._tokenizer(texts))
print("tokenise (host)        %.2f ms" % ms_tok, {k: tuple(v.shape) for k, v in enc.items()})
enc_d = {k: v.to("cuda:0") for k, v in enc.items()}
with torch.no_grad():
    ms_fwd, _ = timed(lambda: p._model(**enc_d).last_hidden_state)
print("forward, one padded batch of 1024  %.2f ms" % ms_fwd)
ms_all, _ = timed(lambda: p.embed_device(texts))
print("embed_device(1024 texts), batch_size %d: %.2f ms" % (b, ms_all))
print("attn impl:", getattr(p._model.config, "_attn_implementation", None))
def enqueue_only(n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        p.embed_device(texts)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t) / n * 1e3, (t2 - t) / n * 1e3
h, w = enqueue_only()
print("embed_device: host returns after %.2f ms per call (enqueue), wall %.2f ms per call; last stats %s" % (h, w, {k: v for k, v in p.last_encode_stats.items() if k != "buckets"}))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(5): p.embed_device(texts)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
