"""Developer: the c5 query encode with the packed forward (_PackedEncoder) against the module forward, same process, alternating."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
texts = synth.query_texts(1024)
prov = {}
for name, packed in (("packed", True), ("module", False)):
    p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=1024)
    p.packed_forward = packed
    prov[name] = p.load()
a = prov["packed"].embed_device(texts); b = prov["module"].embed_device(texts)
torch.cuda.synchronize()
na, nb = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(b, dim=1)
print("max |cos - 1| between the two forwards:", float((1 - (na * nb).sum(1)).abs().max()), " max abs diff:", float((a - b).abs().max()))
for rnd in range(3):
    for name, p in prov.items():
        for _ in range(2): p.embed_device(texts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter(); e0.record()
        for _ in range(10): p.embed_device(texts)
        e1.record(); th = time.perf_counter(); torch.cuda.synchronize(); tw = time.perf_counter()
        print(f"round {rnd} {name:>7}: gpu {e0.elapsed_time(e1)/10:.2f} ms per encode, host returns after {(th-t)/10*1e3:.2f} ms, wall {(tw-t)/10*1e3:.2f} ms", flush=True)
