"""Developer: phase times of k_refine's block 0 (library built with -DRDX_REFINE_STAMPS: tools/ab_lib.py build rstamps "-DRDX_REFINE_STAMPS").
  python tools/refine_stamps.py [rows] [batch] [k]      default: BASELINE config 3 (1 M rows, 256 queries, top-100)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RDX_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librdx_rstamps.so")
import numpy as np, torch
from rag_dpo_amd import _lib, engine, synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ix = engine.HipIndex(1024); ix.reserve(rows)
for j, r0 in enumerate(range(0, rows, synth.CHUNK)):
    ix.add(synth.torch_corpus_chunk(j, min(synth.CHUNK, rows - r0), 1024, "cuda:0"))
q = synth.torch_queries(B, 1024, "cuda:0", total_rows=rows)
s = torch.empty((B, k), dtype=torch.float32, device="cuda"); r = torch.empty((B, k), dtype=torch.int64, device="cuda"); c = torch.empty((B,), dtype=torch.int32, device="cuda")
ix.set_option("profile", 1)
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["segment sizes + prefix", "gather the segments", "k-th largest (radix select)", "build P (coarse >= c_k - 2E)", "exact re-score", "rank + write"]
acc = np.zeros(6)
n = 20
for it in range(n + 5):
    ix.search_device(q, k, s, r, c)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    assert raw.rdx_debug_refine_stamps(st) == 0
    if it >= 5:
        acc += np.diff(np.array([st[i] for i in range(7)], dtype=np.float64)) * 0.01
stt = ix.last_stats()
print(f"rows {rows} B {B} k {k}: k_refine block 0, mean of {n} searches (us); events: refine {stt['ms_refine']*1e3:.1f} us, emitted/query {stt['emitted']/B:.0f}, rescored/query {stt['rescored']/B:.1f}")
for nme, v in zip(names, acc / n):
    print(f"  {nme:32s} {v:7.2f}")
print(f"  {'total':32s} {acc.sum() / n:7.2f}")
