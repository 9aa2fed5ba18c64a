"""Developer A/B of compile-time variants: python tools/ab_lib.py build NAME "-DFOO=1 ..." ; python tools/ab_lib.py run NAME|base [rows]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def path(name): return os.path.join(ROOT, "tools", f"librdx_{name}.so")
if sys.argv[1] == "build":
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *sys.argv[3].split(),
                           os.path.join(ROOT, "rag_dpo_amd", "csrc", "rdx_api.hip"), "-o", path(sys.argv[2])]); sys.exit(0)
which = sys.argv[2]
from rag_dpo_amd import _lib
if which != "base":
    _lib.LIB_PATH = path(which)
import numpy as np
from rag_dpo_amd import engine, synth
corpus = synth.make_corpus(20000, 1024); q = synth.make_queries(300, 1024, corpus)
ix = engine.HipIndex(1024); ix.add(corpus); ix.set_option("force_fast", 1)
s, r, c = ix.search(q, 10)
ix.set_option("force_fast", 0); ix.set_option("force_exact", 1)     # the variant's MFMA path against its own exact full scan
es, er, ec = ix.search(q, 10)
print(which, "parity ids", bool((r == er).all()), "scores", bool((s == es).all()), flush=True)
ix.close()
from tools.quick_bench import build, run
ix = build(int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000)
for b, k in ((1024, 10), (1024, 10), (512, 10)):
    run(ix, b, k, iters=10)
