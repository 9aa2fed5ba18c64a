"""Developer: where the HOST spends the pipelined c5 step (search_begin, next encode's enqueue, search_end), with a cProfile of the encode enqueue."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
from tools.quick_bench import build
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix = build(rows)
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=1024).load()
texts = synth.query_texts(1024)
k = 10
os_ = torch.empty((1024, k), dtype=torch.float32, device="cuda:0"); or_ = torch.empty((1024, k), dtype=torch.int64, device="cuda:0"); oc = torch.empty((1024,), dtype=torch.int32, device="cuda:0")
for _ in range(3):
    q = p.embed_device(texts); ix.search_device(q, k, os_, or_, oc)
torch.cuda.synchronize()
pr = cProfile.Profile()
for mode in ("plain", "profiled"):
    q_next = p.embed_device(texts)
    t = [0.0, 0.0, 0.0]; n = 8
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter()
        ix.search_device_async(q_next, k, os_, or_, oc)
        h1 = time.perf_counter()
        if mode == "profiled": pr.enable()
        q_next = p.embed_device(texts)
        if mode == "profiled": pr.disable()
        h2 = time.perf_counter()
        ix.search_wait()
        h3 = time.perf_counter()
        t[0] += h1 - h0; t[1] += h2 - h1; t[2] += h3 - h2
    torch.cuda.synchronize()
    print(mode, "ms/step %.2f" % ((time.perf_counter() - t0) / n * 1e3), "begin %.2f enqueue %.2f wait %.2f" % tuple(x / n * 1e3 for x in t), flush=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
