"""Developer timing loop (not the contract bench): per-kernel HIP-event times for a few shapes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import engine, synth

def build(n, dim=1024):
    ix = engine.HipIndex(dim)
    ix.reserve(n)
    t = time.time()
    for j, r0 in enumerate(range(0, n, synth.CHUNK)):
        m = min(synth.CHUNK, n - r0)
        ix.add(synth.torch_corpus_chunk(j, m, dim, "cuda:0"))
    torch.cuda.synchronize()
    print(f"built {n} rows in {time.time()-t:.2f}s", flush=True)
    return ix

def run(ix, b, k, iters=5, **opts):
    dim = ix.dim
    for o, v in opts.items():
        ix.set_option(o, v)
    q = synth.torch_queries(b, dim, "cuda:0")
    os_ = torch.empty((b, k), dtype=torch.float32, device="cuda:0")
    or_ = torch.empty((b, k), dtype=torch.int64, device="cuda:0")
    oc = torch.empty((b,), dtype=torch.int32, device="cuda:0")
    ix.set_option("profile", 1)
    for _ in range(2):
        ix.search_device(q, k, os_, or_, oc)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(iters):
        ix.search_device(q, k, os_, or_, oc)
    torch.cuda.synchronize()
    wall = (time.time() - t) / iters * 1e3
    st = ix.last_stats()
    n = len(ix)
    flops = 2.0 * n * b * ix.dim
    ms = st["ms_scan_main"]
    out = {"rows": n, "b": b, "k": k, "wall_ms": round(wall, 3), "qps": round(b / wall * 1e3),
           "scan_ms": round(ms, 3), "scan_TFLOPs": round(flops / ms / 1e9, 1) if ms else None,
           "scan_GBs": round(n * ix.dim * 2 / ms / 1e6, 1) if ms else None,
           **{k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in st.items() if k_.startswith("ms_") or k_ in ("emitted", "rescored", "exact_queries", "path", "sample_rows")}}
    print(json.dumps(out), flush=True)
    for o in opts:
        ix.set_option(o, 0)
    return out

if __name__ == "__main__":
    sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["100000", "1000000"])]
    for n in sizes:
        ix = build(n)
        for b, k in ((1, 50), (4, 50), (64, 10), (256, 100), (1024, 10)):
            run(ix, b, k)
        ix.close()
