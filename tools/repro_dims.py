import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    from rag_dpo_amd import engine
    from oracle import oracle as O
    d, n, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(11)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    ix = engine.HipIndex(d); ix.add(corpus); ix.set_option("force_fast", 1)
    s, r, c = ix.search(q, 10)
    es, er, ec = O.cosine_topk(O.normalize_rows(corpus), q, 10)
    print("d", d, "n", n, "b", b, "ids ok", (r == er).all(), "scores ok", (s == es).all(), ix.last_stats()["exact_queries"], flush=True)
else:
    for d, n, b in ((768, 9000, 70), (200, 9000, 70), (64, 9000, 70), (64, 9000, 40), (128, 3000, 300), (1024, 5000, 33)):
        p = subprocess.run([sys.executable, __file__, str(d), str(n), str(b)], capture_output=True, text=True)
        print("case", d, n, b, "rc", p.returncode, p.stdout.strip()[-200:], p.stderr.strip()[-300:].replace("\n", " | "), flush=True)
