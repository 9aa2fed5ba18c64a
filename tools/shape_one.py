import numpy as np, sys
sys.path.insert(0, sys.argv[1])
from rag_dpo_amd import engine
d, n, b = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
which = sys.argv[5]
rng = np.random.default_rng(11)
corpus = rng.standard_normal((n, d)).astype(np.float32)
q = rng.standard_normal((b, d)).astype(np.float32)
ix = engine.HipIndex(d); ix.add(corpus); ix.set_option("force_fast", 1)
opts = {"none": {}, "fuse": {"fuse_epilogue": 1}, "sib": {"sib_sync": 1}, "cap": {"cand_cap": 8}}[which]
for name, v in opts.items():
    ix.set_option(name, v)
s, r, c = ix.search(q, 10)
allow = np.packbits(np.pad(rng.random(n) < 0.5, (0, (-n) % 32)).reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
ix.search(q, 5, allow)
print("ok", d, n, b, which, ix.last_stats()["exact_queries"], flush=True)
