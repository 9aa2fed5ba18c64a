"""Developer: main-scan time by queries per workgroup (option force_bn) for a few batch sizes: python tools/bn_sweep.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.engine import HipIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ix = HipIndex(1024); ix.reserve(rows)
for j, r0 in enumerate(range(0, rows, synth.CHUNK)):
    ix.add(synth.torch_corpus_chunk(j, min(synth.CHUNK, rows - r0), 1024, "cuda:0"))
ix.set_option("profile", 1)
for b, k in ((256, 100), (192, 10), (256, 10), (384, 10), (512, 10)):
    q = synth.torch_queries(b, 1024, "cuda:0", total_rows=rows)
    s = torch.empty((b, k), dtype=torch.float32, device="cuda:0"); r = torch.empty((b, k), dtype=torch.int64, device="cuda:0"); c = torch.empty((b,), dtype=torch.int32, device="cuda:0")
    base = None
    for bn in (0, 128, 256, 64):
        ix.set_option("force_bn", bn)
        for _ in range(3):
            ix.search_device(q, k, s, r, c)
        ms, tot = 0.0, 0.0
        for _ in range(10):
            ix.search_device(q, k, s, r, c)
            st = ix.last_stats(); ms += st["ms_scan_main"]; tot += st["ms_total"]
        torch.cuda.synchronize()
        if base is None:
            base = (r.clone(), s.clone())
        same = bool((r == base[0]).all() and (s == base[1]).all())
        print(f"rows {rows} B {b} k {k} force_bn {bn}: main scan {ms/10:.4f} ms, search {tot/10:.4f} ms, same result {same}", flush=True)
