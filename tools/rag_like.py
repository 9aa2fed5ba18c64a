"""A corpus shaped like a real RAG store instead of i.i.d. noise: documents of 32 consecutive chunks around a document
centroid, topics of 4096 documents around a topic centroid (so that a query has a few dozen very near rows stored together
and ~1 % of the corpus moderately near). Prints the search time and which path answered."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_dpo_amd import engine, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dim, B, k = 1024, 1024, 10
dev = "cuda:0"
ix = engine.HipIndex(dim); ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(5)
topics = torch.randn((128, dim), generator=g, device=dev)
for j, r0 in enumerate(range(0, rows, synth.CHUNK)):
    m = min(synth.CHUNK, rows - r0)
    docs = torch.randn((m // 32 + 1, dim), generator=g, device=dev) * 0.8 + topics[torch.randint(0, 128, (m // 32 + 1,), generator=g, device=dev)] * 0.6
    ix.add((docs.repeat_interleave(32, dim=0)[:m] + 0.5 * torch.randn((m, dim), generator=g, device=dev)).contiguous())
torch.cuda.synchronize()
rng = np.random.default_rng(1)
src = np.sort(rng.choice(rows, size=B, replace=False))
q = ix.get(src) * np.sqrt(dim) * 0.9 + 0.4 * rng.standard_normal((B, dim)).astype(np.float32)   # a question about that chunk
qd = torch.from_numpy(q.astype(np.float32)).cuda()
s = torch.empty((B, k), dtype=torch.float32, device=dev); r = torch.empty((B, k), dtype=torch.int64, device=dev)
c = torch.empty((B,), dtype=torch.int32, device=dev)
ix.set_option("profile", 1)
for _ in range(3):
    ix.search_device(qd, k, s, r, c)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    ix.search_device(qd, k, s, r, c)
torch.cuda.synchronize()
st = ix.last_stats()
top1 = r[:, 0].cpu().numpy()
print(f"rows {rows}: {(time.perf_counter()-t)/10*1e3:.2f} ms per batch of {B}; main scan {st['ms_scan_main']:.2f} ms; emitted/query {st['emitted']/B:.0f}; "
      f"rescored/query {st['rescored']/B:.1f}; retried {st['retried_queries']}; exact {st['exact_queries']}; "
      f"top-1 is the source chunk for {int((top1 == src).sum())}/{B}, same document for {int((top1 // 32 == src // 32).sum())}/{B}")
