"""What a few 'clustered' queries cost on the 10M-row corpus (rows of one document stored together, missed by the sparse
threshold sample -> candidate overflow): second MFMA pass (default) vs the exact full scan (retry=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.quick_bench import build
from rag_dpo_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix = build(rows)
rng = np.random.default_rng(3)
dim, B, k = 1024, 1024, 10
q = synth.torch_queries(B, dim, "cuda:0").cpu().numpy()
n_cl = 8
for c in range(n_cl):                              # eight clusters of 20 tiles, each between two sampled tiles (stride 64)
    v = rng.standard_normal(dim).astype(np.float32)
    a = (60 * (c + 1) * 64 + 5) * 256
    sigma = rng.uniform(0.2, 0.6, size=(5120, 1)).astype(np.float32)
    ix.update(np.arange(a, a + 5120), v + sigma * rng.standard_normal((5120, dim)).astype(np.float32))
    q[c] = v + 0.1 * rng.standard_normal(dim).astype(np.float32)
qd = torch.from_numpy(q).cuda()
s = torch.empty((B, k), dtype=torch.float32, device="cuda"); r = torch.empty((B, k), dtype=torch.int64, device="cuda")
c_ = torch.empty((B,), dtype=torch.int32, device="cuda")
res = {}
for retry in (1, 0, 1, 0):
    ix.set_option("retry", retry)
    for _ in range(2):
        ix.search_device(qd, k, s, r, c_)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        ix.search_device(qd, k, s, r, c_)
    torch.cuda.synchronize()
    st = ix.last_stats()
    print(f"retry={retry}: {(time.perf_counter()-t)/5*1e3:.2f} ms per batch of {B} ({n_cl} clustered queries); retried {st['retried_queries']}, "
          f"exact {st['exact_queries']}", flush=True)
    res[retry] = r.cpu().numpy().copy()
print("same ids either way:", bool((res[0] == res[1]).all()))
