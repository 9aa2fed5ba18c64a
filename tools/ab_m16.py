"""Developer A/B: same build with -DRDX_MFMA16=1 (v_mfma_f32_16x16x32_f16 scan) vs the product library."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "librdx_m16.so")
if sys.argv[1:] == ["build"]:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DRDX_MFMA16=1",
                           os.path.join(ROOT, "rag_dpo_amd", "csrc", "rdx_api.hip"), "-o", DEV]); sys.exit(0)
which = sys.argv[1]
from rag_dpo_amd import _lib
if which == "m16":
    _lib.LIB_PATH = DEV
import numpy as np
from rag_dpo_amd import engine, synth
from oracle import oracle as O
corpus = synth.make_corpus(20000, 1024); q = synth.make_queries(300, 1024, corpus)
ix = engine.HipIndex(1024); ix.add(corpus); ix.set_option("force_fast", 1)
s, r, c = ix.search(q, 10)
es, er, ec = O.cosine_topk(O.normalize_rows(corpus), q, 10)
print(which, "parity ids", bool((r == er).all()), "scores", bool((s == es).all()), flush=True)
ix.close()
from tools.quick_bench import build, run
ix = build(2_000_000)
for b, k in ((1024, 10), (1024, 10), (64, 10)):
    run(ix, b, k, iters=10)
