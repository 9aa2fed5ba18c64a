"""Developer: rdx_enc_attention_mfma_f16 alone on an ingest-like batch (64 texts, tokens log-normal in [64, 1024]) and on 1024 questions;
TFLOP/s by HIP events (4 * len^2 * 64 flop per text and head), and the same shapes through torch's SDPA on the padded batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_dpo_amd import _lib
L = _lib.load()
heads, H = 16, 1024
rng = np.random.default_rng(3)
def run(lens, name):
    lens = np.asarray(lens, dtype=np.int64); T = int(lens.sum()); B = len(lens)
    first = np.cumsum(lens) - lens
    nb = (lens + 63) // 64
    tix = np.repeat(np.arange(B), nb)
    q0 = (np.arange(int(nb.sum())) - np.repeat(np.cumsum(nb) - nb, nb)) * 64
    qb = torch.from_numpy(np.stack([first[tix], lens[tix], q0, np.zeros_like(q0)], axis=1).astype(np.int32)).cuda()
    qkv = (torch.randn((T, 3 * H), device="cuda") * 1.0).half()
    ctx = torch.empty((T, H), dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: L.rdx_enc_attention_mfma_f16(0, qkv.data_ptr(), qb.data_ptr(), int(qb.shape[0]), heads, 64, 0.125, ctx.data_ptr(), st)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    flops = 4.0 * float((lens.astype(np.float64) ** 2).sum()) * 64 * heads
    # torch SDPA on the padded batch (what the packed forward did for long texts before round 4, without its scatter / gather passes)
    S = int(lens.max())
    qp = torch.randn((B, heads, S, 64), device="cuda", dtype=torch.float16)
    mask = (torch.arange(S, device="cuda")[None, :] < torch.from_numpy(lens).cuda()[:, None]).view(B, 1, 1, S)
    g = lambda: torch.nn.functional.scaled_dot_product_attention(qp, qp, qp, attn_mask=mask)
    for _ in range(3): g()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): g()
    e1.record(); torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / 10
    print(f"{name}: {B} texts, {T} tokens (max {S}): librdx {ms*1e3:.0f} us = {flops / ms / 1e9:.0f} TFLOP/s on real tokens | torch SDPA padded {ms_t*1e3:.0f} us", flush=True)
run(np.clip(np.exp(rng.normal(5.5, 0.65, size=64)), 64, 1024).astype(int), "ingest batch")
run(np.full(64, 1024), "64 x 1024 tokens")
run(np.full(16, 4096), "16 x 4096 tokens")
run(rng.integers(10, 27, size=1024), "1024 questions")
