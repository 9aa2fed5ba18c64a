"""Developer: where the host time of a small search goes (the reference's shape through ShardedSearcher, as bench.py calls it)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.sharded import HipShard, ShardedSearcher
n, dim, b, k = 16919, 1024, 4, 50
sh = HipShard(dim, 0)
sh.add(synth.torch_corpus_chunk(0, n, dim, "cuda:0"))
ss = ShardedSearcher(sh)
q = synth.torch_queries(b, dim, "cuda:0")
for _ in range(50): ss.search(q, k)
t = time.perf_counter()
for _ in range(2000): ss.search(q, k)
print("per search %.1f us" % ((time.perf_counter() - t) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ss.search(q, k)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
