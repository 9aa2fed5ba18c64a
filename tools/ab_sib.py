"""Developer A/B of the sibling lock-step inside ONE process: python tools/ab_sib.py [rows] — alternates no_sib=0/1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rag_dpo_amd import _lib
if os.environ.get("RDX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"librdx_{os.environ['RDX_LIB']}.so")
from rag_dpo_amd import engine, synth
corpus = synth.make_corpus(40000, 1024); q = synth.make_queries(700, 1024, corpus)
ix = engine.HipIndex(1024); ix.add(corpus); ix.set_option("force_fast", 1)
ix.set_option("force_fast", 0); ix.set_option("force_exact", 1)     # reference = the exact full scan of the same library
es, er, ec = ix.search(q, 10)
ix.set_option("force_exact", 0); ix.set_option("force_fast", 1)
for ns in (0, 1):
    ix.set_option("sib_sync", 1 - ns)
    s, r, c = ix.search(q, 10)
    print("no_sib", ns, "parity ids", bool((r == er).all()), "scores", bool((s == es).all()), flush=True)
ix.close()
from tools.quick_bench import build, run
ix = build(int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000)
for rep in range(2):
    for ns in (0, 1):
        for b in (1024, 512):
            o = run(ix, b, 10, iters=10, sib_sync=1 - ns)
            print("   no_sib", ns, "b", b, "scan_ms", o["scan_ms"], "TF", o["scan_TFLOPs"], flush=True)
