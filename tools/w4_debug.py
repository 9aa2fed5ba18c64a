"""Developer: where do the answers of wave_layout = 1 differ from the oracle's? (rows by wave / tile / stream position, queries by tile)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_dpo_amd import engine as eng, synth
from oracle import oracle

rows, dim, b, k = (int(a) for a in (sys.argv[1:5] + ["70000", "1024", "600", "10"][len(sys.argv) - 1:]))
corpus = synth.make_corpus(rows, dim)
q = synth.make_queries(b, dim, corpus)
es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q, k, None)
ix = eng.HipIndex(dim)
ix.add(corpus)
ix.set_option("force_fast", 1)
for wl in (0, 1):
    ix.set_option("wave_layout", wl)
    gs, gr, gc = ix.search(q, k)
    st = ix.last_stats()
    bad = gr != er
    print(f"wave_layout {wl}: mismatched {int(bad.sum())} of {bad.size}; stats {st}", flush=True)
    if bad.any():
        missing = np.array([r for i in range(b) for r in set(er[i].tolist()) - set(gr[i].tolist())], dtype=np.int64)
        qi = np.array([i for i in range(b) for r in set(er[i].tolist()) - set(gr[i].tolist())], dtype=np.int64)
        print("  missing rows:", len(missing), "of", er.size)
        print("  by wave (row % 256 // 64):", np.bincount(missing % 256 // 64, minlength=4).tolist())
        print("  by 16-row block (row % 64 // 16):", np.bincount(missing % 64 // 16, minlength=4).tolist())
        print("  by row % 16 // 4 (lane quad):", np.bincount(missing % 16 // 4, minlength=4).tolist())
        print("  by row % 4 (register):", np.bincount(missing % 4, minlength=4).tolist())
        print("  by query tile:", np.bincount(qi // 256, minlength=3).tolist(), " by query % 16:", np.bincount(qi % 16, minlength=16).tolist())
        print("  by query block (q % 256 // 16):", np.bincount(qi % 256 // 16, minlength=16).tolist())
        tile = missing // 256
        n_tiles = (rows + 255) // 256
        print("  tiles of missing rows: min", tile.min(), "max", tile.max(), "last tile", n_tiles - 1)
        for ns in (64, 80, 128):
            print(f"  if n_streams = {ns}: iteration of the tile in its stream:", np.bincount(tile // ns, minlength=5).tolist(),
                  " last-of-stream share:", float(np.mean(tile + ns >= n_tiles)))
        present = np.array([r for i in range(b) for r in set(er[i].tolist()) & set(gr[i].tolist())], dtype=np.int64)
        print("  present rows by wave:", np.bincount(present % 256 // 64, minlength=4).tolist(), " by tile iteration (80):", np.bincount(present // 256 // 80, minlength=5).tolist())
