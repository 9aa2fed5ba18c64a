"""Developer: phase times of k_select_dense (library built with -DRDX_SELECT_STAMPS: tools/ab_lib.py build stamps "-DRDX_SELECT_STAMPS")."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RDX_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librdx_stamps.so")
import numpy as np
from rag_dpo_amd import _lib, engine, synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16919
corpus = synth.make_corpus(rows, 1024); q = synth.make_queries(4, 1024, corpus)
ix = engine.HipIndex(1024); ix.add(corpus)
for _ in range(5):
    ix.search(q, 50)
raw = ctypes.CDLL(_lib.LIB_PATH)
st = (ctypes.c_ulonglong * 16)()
assert raw.rdx_debug_select_stamps(st) == 0
t = [st[i] for i in range(7)]
names = ["load keys", "radix select (4 passes)", "setup", "collect pass", "equal-key ranks", "count finite", "rank_and_write"]
for i, n in enumerate(names[:6] + names[6:]):
    if i + 1 < 7:
        print(f"{names[i]:28s} {(t[i + 1] - t[i]) * 10:8d} ns")
print("total", (t[6] - t[0]) * 10, "ns")
