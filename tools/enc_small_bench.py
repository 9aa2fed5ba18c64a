"""Developer: latency of the single-question forward (embed_device([q]) + synchronise, HIP-graph replay) under the knobs of
rag_dpo_amd/embedding_provider.py `_PackedEncoder` (RDX_ENC_FPB_O, RDX_ENC_FPB_F2). One JSON line.
  python tools/enc_small_bench.py [reps]         RDX_ENC_OLD=1: round 3's seven-launches-per-layer path for comparison"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider, _PackedEncoder
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get("RDX_ENC_OLD"):
    _PackedEncoder.STAGE_TOKENS = 0
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=64).load()
texts = synth.query_texts(64)
tk = lambda t: len(t.split()) + 2
short, mid, longq = [t for t in texts if tk(t) <= 16][:4], [t for t in texts if 17 <= tk(t) <= 24][:6], [t for t in texts if 25 <= tk(t) <= 32][:6]
out = {"fpb_o": p._packed.stage_fpb_o, "fpb_f2": p._packed.stage_fpb_f2,
       "old_path": bool(os.environ.get("RDX_ENC_OLD"))}
for name, sample in (("q25_32_tokens", longq), ("q17_24_tokens", mid), ("q_le16_tokens", short)):
    if not sample:
        continue
    for t in sample:
        for _ in range(3): p.embed_device([t])
    torch.cuda.synchronize()
    lat = []
    for rep in range(reps):
        for t in sample:
            t0 = time.perf_counter(); p.embed_device([t]); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    # GPU time of the replay alone: events around 20 back-to-back replays of one question
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): p.embed_device([sample[0]])
    e1.record(); torch.cuda.synchronize()
    out[name] = {"median_ms": round(lat[len(lat) // 2], 4), "min_ms": round(lat[0], 4), "p90_ms": round(lat[int(len(lat) * 0.9)], 4),
                 "back_to_back_ms": round(e0.elapsed_time(e1) / 20, 4), "tokens": [len(t.split()) + 2 for t in sample]}
print(json.dumps(out), flush=True)
