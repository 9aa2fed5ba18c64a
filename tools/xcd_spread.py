"""Developer: the XCD finish spread and shares of the main scan, search by search (10 M x 1024, B = 1024): noise or oscillation?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from tools.quick_bench import build
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix = build(n)
b, k = 1024, 10
os_ = torch.empty((b, k), dtype=torch.float32, device="cuda:0"); or_ = torch.empty((b, k), dtype=torch.int64, device="cuda:0")
oc = torch.empty((b,), dtype=torch.int32, device="cuda:0")
ix.set_option("profile", 2)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    q = synth.torch_queries(b, 1024, "cuda:0", seed=100 + i) if "seed" in synth.torch_queries.__code__.co_varnames else synth.torch_queries(b, 1024, "cuda:0")
    ix.search_device(q, k, os_, or_, oc)
    torch.cuda.synchronize()
    st = ix.last_stats()
    print(i, round(st["ms_scan_main"], 3), {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in st.items() if "xcd" in k_}, flush=True)
