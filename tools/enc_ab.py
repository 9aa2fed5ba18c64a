"""Developer: the c5 query encode with the packed forward (_PackedEncoder; with librdx's encoder kernels and on torch operations) against the module forward, same process, alternating."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
texts = synth.query_texts(1024)
prov = {}
import os
variants = [("fused", True, None, None), ("packed", True, False, None), ("module", False, False, None)]
if os.environ.get("ENC_AB_GRAPH"):
    variants.insert(1, ("graph", True, None, True))
for name, packed, fused, graphs in variants:
    p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=1024)
    p.packed_forward, p.fused_kernels, p.encoder_graphs = packed, fused, graphs
    prov[name] = p.load()
b = prov["module"].embed_device(texts)
for name in [v[0] for v in variants if v[0] != "module"]:
    prov[name].embed_device(texts); prov[name].embed_device(texts)      # (a graph is captured the second time a shape is seen)
    a = prov[name].embed_device(texts)
    torch.cuda.synchronize()
    na, nb = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(b, dim=1)
    print(name, "vs module: max |cos - 1|", float((1 - (na * nb).sum(1)).abs().max()), " max abs diff:", float((a - b).abs().max()))
# GPU time of ONE encode with the host out of the picture: a plug of ~60 ms of GEMMs is enqueued first, the encode is enqueued while
# the plug runs (a busy box's host needs 30 - 60 ms for the encode's launches), events bracket the encode on the stream.
big = torch.randn((8192, 8192), dtype=torch.float16, device="cuda:0")
def plug(n=60):
    for _ in range(n):
        torch.mm(big, big)
for rnd in range(3):
    for name, p in prov.items():
        for _ in range(2): p.embed_device(texts)
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            plug(); e0.record(); t = time.perf_counter(); p.embed_device(texts); th = time.perf_counter(); e1.record()
            torch.cuda.synchronize(); ms.append((e0.elapsed_time(e1), (th - t) * 1e3))
        print(f"round {rnd} {name:>7}: GPU time per encode behind a plug " + " ".join(f"{a:.2f}" for a, _ in ms) + "  ms; host enqueue " + " ".join(f"{b:.1f}" for _, b in ms), flush=True)
