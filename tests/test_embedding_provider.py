"""EmbeddingProvider mirror (reference src/utils/embedding_provider.py:34-191): interface on CPU, numerics on GPU."""
import inspect

import numpy as np
import pytest

from rag_dpo_amd import embedding_provider as EP


def test_interface_mirrors_reference():
    assert (EP.DEFAULT_MODEL, EP.DEFAULT_DIMS, EP.DEFAULT_BATCH_SIZE, EP.MAX_SEQ_LENGTH, EP.TRUNCATE_CHARS) == \
        ("BAAI/bge-m3", 1024, 64, 8192, 20000)                     # reference :25-31
    sig = inspect.signature(EP.EmbeddingProvider.__init__)
    assert list(sig.parameters)[1:] == ["model_name", "device", "dtype", "batch_size", "cache_dir"]   # reference :44-51
    p = EP.EmbeddingProvider(cache_dir="/nonexistent")
    assert p.model_name == "BAAI/bge-m3" and p.dims == 1024 and p.is_loaded is False
    assert p.embed([]) == []                                                          # reference :128-129
    for name in ("load", "unload", "embed", "embed_query", "is_available", "get_info"):
        assert callable(getattr(p, name))
    info = p.get_info()
    assert set(info) == {"model", "device", "dtype", "dims", "loaded", "vram_gb", "batch_size"}   # reference :170-181
    assert "not loaded" in repr(p)
    with pytest.raises(RuntimeError, match="never downloads"):
        p.embed(["bonjour"])                       # by-name fetch is unavailable offline: loud failure, no fallback


@pytest.mark.gpu
def test_embed_unit_norm_and_k1_parity(oracle):
    p = EP.EmbeddingProvider(model_name="random-init:tiny", device="cuda")
    texts = ["Comment faire une AIPD ?", "durée de conservation", "x " * 30000, ""]
    out = p.embed(texts)
    assert len(out) == 4 and all(len(v) == p.dims == 64 for v in out) and isinstance(out[0][0], float)
    a = np.asarray(out, dtype=np.float64)
    assert np.abs(np.linalg.norm(a, axis=1) - 1).max() < 1e-6
    raw = p._encode_raw([t[:EP.TRUNCATE_CHARS] for t in texts]).cpu().numpy()
    np.testing.assert_array_equal(np.asarray(out, dtype=np.float32), oracle.normalize_rows(raw))   # K1 == oracle, bit for bit
    assert p.embed_query(texts[0]) == out[0] and p.is_loaded
    p.unload()
    assert not p.is_loaded


@pytest.mark.gpu
def test_embed_device_feeds_the_index_without_the_list_round_trip(oracle):
    """embed_device() hands the raw CLS embeddings to the device index directly; the index normalises them with the same
    K1 — so the stored rows equal the embed() -> lists -> add() route up to that route's second normalisation (1 ulp) and
    the neighbours are the same"""
    from rag_dpo_amd.engine import HipIndex
    p = EP.EmbeddingProvider(model_name="random-init:tiny", device="cuda").load()   # dims are known once the model is loaded
    texts = [f"chunk numéro {i} sur la durée de conservation" for i in range(300)]
    a, b = HipIndex(p.dims), HipIndex(p.dims)
    a.add(np.asarray(p.embed(texts), dtype=np.float32))
    b.add(p.embed_device(texts))
    rows = np.arange(300)
    np.testing.assert_allclose(a.get(rows), b.get(rows), rtol=0, atol=2.5e-7)
    np.testing.assert_array_equal(b.get(rows), oracle.normalize_rows(p.embed_device(texts).cpu().numpy()))   # one K1, bit for bit
    q = np.asarray(p.embed(["durée de conservation ?"]), dtype=np.float32)
    sa, sb = a.search(q, 5), b.search(q, 5)
    np.testing.assert_array_equal(sa[1], sb[1])
    assert p.embed_device([]).shape == (0, p.dims)


# ---- the real-checkpoint branch of load(): a LOCAL directory with config + weights + tokenizer files -----------------------
# No BGE-M3 weights exist offline, so the checkpoint is built here: a sentencepiece unigram model trained on a small synthetic
# corpus, wrapped as an XLM-RoBERTa tokenizer, and a 2-layer XLM-R with random weights, both written with save_pretrained —
# the directory layout `BAAI/bge-m3` has. What this pins is the PIPELINE (char-truncate -> HF tokenizer -> XLM-R forward -> CLS
# pooling -> L2 normalise, reference src/utils/embedding_provider.py:118-147) against a plain fp32 transformers forward of the
# same checkpoint; the VALUES of BGE-M3 itself stay unpinned.

def make_local_checkpoint(path):
    import random
    import sentencepiece as spm
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaModel, XLMRobertaTokenizer
    os_ = __import__("os")
    os_.makedirs(path, exist_ok=True)
    words = ("durée conservation données personnelles traitement registre sous-traitant responsable AIPD analyse impact "
             "consentement cookies traceurs vidéosurveillance salariés transfert hors union européenne violation notification "
             "CNIL délégué protection base légale intérêt légitime droit accès effacement portabilité sanction mise en demeure "
             "sécurité Quelle Comment des de la le une faire").split()
    rnd = random.Random(0)
    txt = os_.path.join(path, "corpus.txt")
    with open(txt, "w", encoding="utf-8") as f:
        for _ in range(3000):
            f.write(" ".join(rnd.choice(words) for _ in range(rnd.randint(5, 30))) + " ?\n")
    spm.SentencePieceTrainer.train(input=txt, model_prefix=os_.path.join(path, "spm"), vocab_size=150, model_type="unigram",
                                   character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    sp = spm.SentencePieceProcessor(model_file=os_.path.join(path, "spm.model"))
    pieces = [(sp.id_to_piece(i), sp.get_score(i)) for i in range(sp.get_piece_size()) if not (sp.is_control(i) or sp.is_unknown(i))]
    vocab = [("<s>", 0.0), ("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0)] + pieces + [("<mask>", 0.0)]   # XLM-R's id layout
    tok = XLMRobertaTokenizer(vocab=vocab)
    ckpt = os_.path.join(path, "bge-m3-local")
    tok.save_pretrained(ckpt)
    torch.manual_seed(0)
    cfg = XLMRobertaConfig(vocab_size=len(vocab), hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                           max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    XLMRobertaModel(cfg, add_pooling_layer=False).save_pretrained(ckpt)
    return ckpt


TEXTS = ["Comment faire une AIPD ?", "durée de conservation des données personnelles", "registre", "",
         "Quelle sanction la CNIL ? " * 40, "transfert hors union européenne", "cookies traceurs consentement"]


def hf_reference(ckpt, texts):
    """plain transformers, fp32, CPU: tokenizer -> forward -> CLS, one text at a time (no padding involved)"""
    import torch
    from transformers import AutoTokenizer, XLMRobertaModel
    tok = AutoTokenizer.from_pretrained(ckpt, local_files_only=True)
    model = XLMRobertaModel.from_pretrained(ckpt, add_pooling_layer=False, local_files_only=True).eval()
    out = []
    with torch.no_grad():
        for t in texts:
            enc = tok([t], padding=True, truncation=True, max_length=EP.MAX_SEQ_LENGTH, return_tensors="pt")
            out.append(model(**enc).last_hidden_state[0, 0].to(torch.float32).numpy())
    return np.stack(out), tok


def test_local_checkpoint_branch_cpu(tmp_path):
    import torch
    ckpt = make_local_checkpoint(str(tmp_path))
    ref, tok = hf_reference(ckpt, TEXTS)
    ids = tok(TEXTS[1])["input_ids"]
    assert len(ids) > 4 and tok.unk_token_id not in ids                    # a real subword tokenizer, not the hashing stand-in
    # model_name = the directory itself; and <cache_dir>/<name> as the reference passes cache_dir (embedding_provider.py:87-92)
    for kw in (dict(model_name=ckpt), dict(model_name="bge-m3-local", cache_dir=str(tmp_path))):
        p = EP.EmbeddingProvider(device="cpu", dtype=torch.float32, batch_size=3, **kw)
        assert p.load() is p and p.is_loaded and p.dims == 64
        assert not isinstance(p._tokenizer, EP._HashTokenizer)
        raw = p._encode_raw(TEXTS).numpy()                                  # length-sorted batches of 3, padded, back in input order
        np.testing.assert_allclose(raw, ref, rtol=0, atol=2e-5)            # padding + batching change nothing beyond fp32 noise
        assert p.embed_device(TEXTS[:2]).shape == (2, 64)
        with pytest.raises(Exception, match="librdx|MI355X|GPU"):
            p.embed(TEXTS)                                                  # the normalise step runs in librdx: no CPU path
        p.unload()


@pytest.mark.gpu
def test_local_checkpoint_embed_matches_fp32_reference(tmp_path, oracle):
    """embed() on the GPU (fp16 forward, K1 normalise) against the fp32 transformers forward + oracle normalise of the same local
    checkpoint: unit rows, cosine to the reference row > 0.9999, max abs difference 3e-3 (fp16 forward of a 2-layer model)"""
    import torch
    ckpt = make_local_checkpoint(str(tmp_path))
    ref, _ = hf_reference(ckpt, TEXTS)
    ref_hat = oracle.normalize_rows(ref)
    p = EP.EmbeddingProvider(model_name=ckpt, device="cuda", dtype=torch.float16, batch_size=4)
    out = np.asarray(p.embed(TEXTS), dtype=np.float32)
    assert out.shape == (len(TEXTS), 64)
    assert np.abs(np.linalg.norm(out.astype(np.float64), axis=1) - 1).max() < 1e-6
    assert (np.sum(out * ref_hat, axis=1) > 0.9999).all()
    assert np.abs(out - ref_hat).max() < 3e-3
    p32 = EP.EmbeddingProvider(model_name=ckpt, device="cuda", dtype=torch.float32, batch_size=4)
    out32 = np.asarray(p32.embed(TEXTS), dtype=np.float32)
    assert np.abs(out32 - ref_hat).max() < 2e-5                             # fp32 on the GPU: the same pipeline to fp32 noise


def test_length_buckets_change_padding_not_values():
    """embed_device cuts a batch into token-count buckets (at most max_buckets forwards, each at its own width): fewer padding
    tokens, the same CLS rows as ONE forward padded to the longest text (attention masks hide the padding either way)."""
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    rng = np.random.default_rng(3)
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(3, 60, size=300)]
    texts[17] = " ".join(rng.choice(words, size=200))              # one long outlier: the reason one width is wasteful
    p = EmbeddingProvider(model_name="random-init:tiny", device="cpu", dtype=torch.float32, batch_size=512)
    p.packed_forward = False                                      # (the module-by-module forward: the one that pays for padding)
    p.load()
    p.max_buckets, p.bucket_granule, p.bucket_overhead_tokens = 1, 64, 0
    one = p._encode_raw(texts)
    s1 = dict(p.last_encode_stats)
    assert len(s1["buckets"]) == 1 and s1["tokens_padded"] == s1["tokens_padded_one_width"] == 300 * 202
    p.max_buckets, p.bucket_overhead_tokens = 4, 64
    four = p._encode_raw(texts)
    s4 = p.last_encode_stats
    assert 2 <= len(s4["buckets"]) <= 4 and sum(b["rows"] for b in s4["buckets"]) == 300
    assert s4["tokens_real"] == s1["tokens_real"] and s4["tokens_padded"] < 0.5 * s1["tokens_padded"]
    widths = [b["width"] for b in s4["buckets"]]
    assert widths == sorted(widths, reverse=True) and widths[0] == 202
    assert torch.allclose(one, four, atol=2e-5, rtol=1e-5)
    # the cut chooser: optimal on a case small enough to enumerate
    lens = np.array(sorted(rng.integers(3, 100, size=256).tolist(), reverse=True))
    p.max_buckets, p.bucket_granule, p.bucket_overhead_tokens = 3, 64, 500
    cuts = p._bucket_cuts(lens)
    cost = lambda c: sum((b - a) * lens[a] + 500 for a, b in zip(c[:-1], c[1:]))
    cands = [[0, 256]] + [[0, i, 256] for i in (64, 128, 192)] + [[0, i, j, 256] for i in (64, 128) for j in (128, 192) if j > i]
    assert cuts[0] == 0 and cuts[-1] == 256 and cost(cuts) == min(cost(c) for c in cands)


def test_packed_forward_equals_the_module_forward():
    """_PackedEncoder (token-wise layers over the real tokens only, Q/K/V scattered into the padded layout just for the attention,
    one GEMM for the three projections) against `XLMRobertaModel.forward` on the same weights: the same CLS rows, texts of 1 to
    200 tokens in one batch, and through embed_device()'s batching"""
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    rng = np.random.default_rng(9)
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(1, 60, size=120)] + [" ".join(rng.choice(words, size=200)), "w1"]
    ref = EmbeddingProvider(model_name="random-init:tiny", device="cpu", dtype=torch.float32, batch_size=512)
    ref.packed_forward = False
    ref.load()
    fast = EmbeddingProvider(model_name="random-init:tiny", device="cpu", dtype=torch.float32, batch_size=512).load()
    assert fast._packed is not None and ref._packed is None
    a, b = ref._encode_raw(texts), fast._encode_raw(texts)
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-5), float((a - b).abs().max())
    assert len(fast.last_encode_stats["buckets"]) == 1            # padding costs the attention only: one forward
    fast.batch_size = 50                                          # several batches
    assert torch.allclose(fast._encode_raw(texts), a, atol=2e-5, rtol=1e-5)


def _attention_reference(qkv, lens, heads):
    """softmax(q k^T / sqrt(d)) v per text and head in fp32, on the packed [T][3H] projection"""
    import torch
    T, H3 = qkv.shape
    H = H3 // 3
    d = H // heads
    out = torch.empty((T, H), dtype=torch.float32)
    t0 = 0
    x = qkv.float()
    for n in lens:
        blk = x[t0:t0 + n]
        q, k, v = (blk[:, i * H:(i + 1) * H].view(n, heads, d).transpose(0, 1) for i in range(3))   # [heads][n][d]
        p = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, dim=-1)
        out[t0:t0 + n] = (p @ v).transpose(0, 1).reshape(n, H)
        t0 += n
    return out


@pytest.mark.gpu
def test_encoder_attention_kernel_vs_fp32_reference():
    """rdx_enc_attention_f16 (self-attention of short texts on the packed QKV projection) against a plain fp32 torch reference of the
    same op: ragged lengths 1..64, a token count that is no multiple of the block's 64 rows, large score magnitudes.
    Tolerance: the output is fp16 (half an ulp of values up to ~4: 2e-3) on fp32 arithmetic over the same fp16 inputs."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    # (the last shape: texts longer than the kernel's LDS window of keys — those rows read global memory, same arithmetic)
    for heads, lens in ((16, [1, 64, 7, 20, 33, 2, 19, 21, 5]), (8, list(rng.integers(1, 30, size=200))), (2, [3]), (2, [100, 150, 30, 200, 1, 64])):
        H = heads * 64
        T = int(sum(lens))
        g = torch.Generator().manual_seed(T)
        qkv = (torch.randn((T, 3 * H), generator=g) * 1.5).half()
        first = np.cumsum(lens) - np.asarray(lens)
        tf = torch.from_numpy(np.repeat(first, lens).astype(np.int32)).cuda()
        tl = torch.from_numpy(np.repeat(lens, lens).astype(np.int32)).cuda()
        qd = qkv.cuda()
        ctx = torch.full((T, H), float("nan"), dtype=torch.float16, device="cuda")
        want = _attention_reference(qkv, [int(n) for n in lens], heads)
        # max_text_tokens sizes the LDS window: the true maximum, unknown (0), and a too small promise (costs speed only)
        for promise in (int(max(lens)), 0, 2):
            ctx.fill_(float("nan"))
            rc = L.rdx_enc_attention_f16(0, qd.data_ptr(), tf.data_ptr(), tl.data_ptr(), T, heads, 64, 0.125, promise, ctx.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
            assert rc == 0, _lib.last_error()
            torch.cuda.synchronize()
            got = ctx.float().cpu()
            assert torch.isfinite(got).all()
            assert float((got - want).abs().max()) <= 3e-3, (promise, float((got - want).abs().max()))
    # argument checks: wrong head width, misaligned pointer
    assert L.rdx_enc_attention_f16(0, qd.data_ptr(), tf.data_ptr(), tl.data_ptr(), T, heads, 32, 0.125, 0, ctx.data_ptr(), 0) != 0
    assert L.rdx_enc_attention_f16(0, qd.data_ptr() + 2, tf.data_ptr(), tl.data_ptr(), T, heads, 64, 0.125, 0, ctx.data_ptr(), 0) != 0
    assert L.rdx_enc_attention_f16(0, qd.data_ptr(), tf.data_ptr(), tl.data_ptr(), 0, heads, 64, 0.125, 0, ctx.data_ptr(), 0) == 0


@pytest.mark.gpu
def test_encoder_add_layernorm_kernel_vs_torch():
    """rdx_enc_add_layernorm_f16 against torch's fp16 add followed by LayerNorm (fp32 statistics) and against the fp32 reference of
    LayerNorm(half(a + b)): every supported width, a row count that is no multiple of 4. Tolerance: one fp16 ulp of the output."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    for hidden in (512, 1024, 1536, 2048):
        g = torch.Generator().manual_seed(hidden)
        rows = 1001
        a = (torch.randn((rows, hidden), generator=g) * 2).half().cuda()
        b = (torch.randn((rows, hidden), generator=g) + 0.5).half().cuda()
        ln = torch.nn.LayerNorm(hidden, eps=1e-5)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(hidden, generator=g) * 0.3 + 1)
            ln.bias.copy_(torch.randn(hidden, generator=g) * 0.2)
        ref32 = ln((a + b).float().cpu())                        # fp32 LayerNorm of the fp16-rounded sum
        ln = ln.half().cuda()
        out = torch.empty_like(a)
        rc = L.rdx_enc_add_layernorm_f16(0, a.data_ptr(), b.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), 1e-5, rows, hidden,
                                         out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, _lib.last_error()
        with torch.no_grad():
            want = ln(a + b)
        torch.cuda.synchronize()
        ulp = 2.0 ** -10 * torch.clamp(want.float().abs(), min=1.0)
        assert bool(((out.float() - want.float()).abs() <= ulp).all()), float((out.float() - want.float()).abs().max())
        ref16 = torch.nn.functional.layer_norm((a + b).float().cpu(), (hidden,), ln.weight.float().cpu(), ln.bias.float().cpu(), 1e-5)
        assert float((out.float().cpu() - ref16.detach()).abs().max()) <= 4e-3
        del ref32
    assert L.rdx_enc_add_layernorm_f16(0, a.data_ptr(), b.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), 1e-5, rows, 768,
                                       out.data_ptr(), 0) != 0


@pytest.mark.gpu
def test_encoder_small_linear_kernel_vs_fp32_reference():
    """rdx_enc_linear_small_f16 (the projections of a forward over at most 256 tokens: the weight matrix streamed once, K split over a
    workgroup's waves) against a plain torch fp32 reference of the same op, with and without the erf GELU: token counts 1..256 that are
    no multiples of 16, the encoder's four shapes. Tolerance: fp16 output of fp32 accumulation over fp16 inputs (half an ulp relative
    + the reference's own rounding): 2e-3 relative to the row's scale."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(12)
    for T, N, K in ((1, 1024, 1024), (20, 3072, 1024), (32, 4096, 1024), (45, 1024, 4096), (128, 1024, 1024), (200, 4096, 1024), (256, 1024, 4096), (7, 512, 512)):
        x = (torch.randn((T, K), generator=g)).half().cuda()
        w = (torch.randn((N, K), generator=g) * K ** -0.5).half().cuda()
        b = (torch.randn((N,), generator=g) * 0.1).half().cuda()
        for act in (0, 1):
            out = torch.full((T, N), float("nan"), dtype=torch.float16, device="cuda")
            rc = L.rdx_enc_linear_small_f16(0, x.data_ptr(), w.data_ptr(), b.data_ptr(), T, N, K, act, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, _lib.last_error()
            want = x.float() @ w.float().T + b.float()
            if act:
                want = torch.nn.functional.gelu(want)
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            err = (out.float() - want).abs().max()
            assert float(err) <= 2e-3 * max(1.0, float(want.abs().max())), (T, N, K, act, float(err))
    assert L.rdx_enc_linear_small_f16(0, x.data_ptr(), w.data_ptr(), b.data_ptr(), 257, N, K, 0, out.data_ptr(), 0) != 0
    assert L.rdx_enc_linear_small_f16(0, x.data_ptr(), w.data_ptr(), b.data_ptr(), T, 24, K, 0, out.data_ptr(), 0) != 0
    assert L.rdx_enc_linear_small_f16(0, x.data_ptr(), w.data_ptr(), b.data_ptr(), T, N, 768, 0, out.data_ptr(), 0) != 0


@pytest.mark.gpu
def test_fused_encoder_kernels_equal_the_torch_operations():
    """the whole packed forward with librdx's attention and add + LayerNorm kernels against the same forward on torch operations
    (and against the module forward): same weights (seeded), fp16 on the GPU, 64-wide heads; texts of 1..60 tokens take the kernels,
    a batch with one long text takes the torch path — both must agree with the reference to fp16 accuracy"""
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    rng = np.random.default_rng(4)
    words = [f"w{i}" for i in range(300)]
    short = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(1, 40, size=300)] + ["w1"]
    long_ = short[:20] + [" ".join(rng.choice(words, size=150))]

    def make(fused, packed=True):
        p = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=512)
        p.fused_kernels, p.packed_forward = fused, packed
        return p.load()

    fast, plain, module = make(None), make(False), make(False, packed=False)
    assert fast._packed.fused and not plain._packed.fused and module._packed is None
    replayed = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=512)
    replayed.encoder_graphs = True                         # the fused forward as a HIP graph: eager, capture + replay, replay
    replayed.load()
    want = fast.embed_device(short).clone()
    runs = [replayed.embed_device(short).clone() for _ in range(3)]
    assert len(replayed._packed._graph) == 1
    for r in runs:
        assert float((torch.nn.functional.normalize(r, dim=1) - torch.nn.functional.normalize(want, dim=1)).abs().max()) <= 2e-3
    assert torch.equal(runs[1], runs[2])
    # the default (graphs = "auto"): batches of up to 8 texts are replayed from a canonical padded shape — embed_query() and a question's
    # <= 4 sub-queries, the reference's online path. Different questions of similar size share ONE graph; values equal the eager forward's
    assert fast._packed.graphs == "auto" and len(fast._packed._graph) == 0
    small = [["w9 w8 w7 w6 w5"], ["w1 w2 w3"], ["w4 w5 w6 w7", "w1", "w2 w3 w4 w5 w6 w7 w8"], ["w9 w9", "w3 w1 w2"]]
    for texts in small + small:                             # second round: every shape has been seen -> captured or replayed
        got = fast.embed_device(texts).clone()
        ref = plain.embed_device(texts)
        assert got.shape == ref.shape
        assert float((torch.nn.functional.normalize(got, dim=1) - torch.nn.functional.normalize(ref, dim=1)).abs().max()) <= 2e-3, texts
    assert len(fast._packed._graph) == 3                    # three canonical shapes: 8, 16 and 24 tokens (multiples of 8 up to 32)
    a1, a2 = fast.embed_device(small[2]).clone(), fast.embed_device(small[2]).clone()
    assert torch.equal(a1, a2)
    many = [" ".join(f"w{i}" for i in range(j, j + 30)) for j in range(4)]      # 4 x 32 tokens: another canonical shape
    for _ in range(3):
        got = fast.embed_device(many).clone()
    assert len(fast._packed._graph) == 4
    assert float((torch.nn.functional.normalize(got, dim=1) - torch.nn.functional.normalize(plain.embed_device(many), dim=1)).abs().max()) <= 2e-3
    fast._packed.MAX_GRAPHS = 1                             # capturing a new shape pushes the least recently used ones out
    for _ in range(2):
        fast.embed_device(many + many[:2])                  # 6 x 32 tokens: a third canonical shape
    assert list(fast._packed._graph) == [("small", 192, 32)]




@pytest.mark.gpu
def test_encoder_stage_kernel_vs_fp32_reference():
    """rdx_enc_stage_f16 (one projection of the single-question forward: LayerNorm prologue or plain / gathered input, the weight matrix
    streamed once by workgroups of 16 / 8 / 4 features, bias / erf GELU / residual epilogue) against a plain torch fp32 reference of the
    same op on the same fp16 inputs. Token counts 1..32 on both sides of the 16-token block, every K the kernel takes, with and without
    several input widths. Tolerance: fp16 output of fp32 accumulation (2e-3 relative to the row scale, as for
    rdx_enc_linear_small_f16); the stored LayerNorm output within one fp16 ulp of torch's fp32 LayerNorm."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(21)
    # LayerNorm prologue: qkv / FFN-up shapes
    for T, N, K, epi in ((1, 3072, 1024, 0), (16, 4096, 1024, 1), (17, 3072, 1024, 0), (20, 4096, 1024, 1), (32, 3072, 1024, 0), (32, 1024, 512, 1), (9, 1536, 512, 0)):
        s = (torch.randn((T, K), generator=g) * 1.7 + 0.3).half().cuda()
        w = (torch.randn((N, K), generator=g) * K ** -0.5).half().cuda()
        b = (torch.randn((N,), generator=g) * 0.1).half().cuda()
        ln = torch.nn.LayerNorm(K, eps=1e-5)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(K, generator=g) * 0.3 + 1)
            ln.bias.copy_(torch.randn(K, generator=g) * 0.2)
        ln = ln.half().cuda()
        y_ref = torch.nn.functional.layer_norm(s.float(), (K,), ln.weight.float(), ln.bias.float(), 1e-5)
        for rep in range(2):
            out = torch.full((T, N), float("nan"), dtype=torch.float16, device="cuda")
            y = torch.full((T, K), float("nan"), dtype=torch.float16, device="cuda")
            rc = L.rdx_enc_stage_f16(0, s.data_ptr(), None, ln.weight.data_ptr(), ln.bias.data_ptr(), 1e-5, y.data_ptr(), w.data_ptr(), b.data_ptr(),
                                     None, T, N, K, epi, 0, out.data_ptr(), st)
            assert rc == 0, _lib.last_error()
            torch.cuda.synchronize()
            assert torch.isfinite(out).all() and torch.isfinite(y).all()
            ulp = 2.0 ** -10 * torch.clamp(y_ref.abs(), min=1.0)
            assert bool(((y.float() - y_ref).abs() <= ulp).all()), (T, N, K, float((y.float() - y_ref).abs().max()))
            want = y.float() @ w.float().T + b.float()            # (from the fp16 LayerNorm output the kernel itself multiplies)
            if epi:
                want = torch.nn.functional.gelu(want)
            err = float((out.float() - want).abs().max())
            assert err <= 2e-3 * max(1.0, float(want.abs().max())), (T, N, K, epi, err)
    # plain input: output projection / FFN-down shapes, residual epilogue, 16 / 8 / 4 features per workgroup, gathered rows
    for T, N, K in ((1, 1024, 1024), (16, 1024, 4096), (20, 1024, 1024), (32, 1024, 4096), (27, 512, 1024), (32, 512, 2048), (5, 512, 512)):
        rows_src = 40
        x = torch.randn((rows_src, K), generator=g).half().cuda()
        res = torch.randn((rows_src, N), generator=g).half().cuda()
        w = (torch.randn((N, K), generator=g) * K ** -0.5).half().cuda()
        b = (torch.randn((N,), generator=g) * 0.1).half().cuda()
        idx = torch.randperm(rows_src, generator=g)[:T].cuda()
        for fpb in (16, 8, 4):
            for epi in (0, 1, 2):
                for gather in (False, True):
                    out = torch.full((T, N), float("nan"), dtype=torch.float16, device="cuda")
                    rc = L.rdx_enc_stage_f16(0, x.data_ptr(), idx.data_ptr() if gather else None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(),
                                             res.data_ptr(), T, N, K, epi, fpb, out.data_ptr(), st)
                    assert rc == 0, _lib.last_error()
                    xs, rs = (x[idx], res[idx]) if gather else (x[:T], res[:T])
                    want = xs.float() @ w.float().T + b.float()
                    if epi == 1:
                        want = torch.nn.functional.gelu(want)
                    if epi == 2:
                        want = want.half().float() + rs.float()
                    torch.cuda.synchronize()
                    assert torch.isfinite(out).all()
                    err = float((out.float() - want).abs().max())
                    assert err <= 3e-3 * max(1.0, float(want.abs().max())), (T, N, K, fpb, epi, gather, err)
    # argument checks
    assert L.rdx_enc_stage_f16(0, x.data_ptr(), None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(), None, 33, N, K, 0, 16, out.data_ptr(), 0) != 0
    assert L.rdx_enc_stage_f16(0, x.data_ptr(), None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(), None, 4, N, 768, 0, 16, out.data_ptr(), 0) != 0
    assert L.rdx_enc_stage_f16(0, x.data_ptr(), None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(), None, 4, N, K, 2, 16, out.data_ptr(), 0) != 0
    assert L.rdx_enc_stage_f16(0, x.data_ptr(), None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(), None, 4, N, K, 0, 5, out.data_ptr(), 0) != 0
    assert L.rdx_enc_stage_f16(0, x.data_ptr(), None, None, None, 0.0, None, w.data_ptr(), b.data_ptr(), None, 0, N, K, 0, 16, out.data_ptr(), 0) == 0


@pytest.mark.gpu
def test_encoder_attention_small_kernel_vs_fp32_reference():
    """rdx_enc_attention_small_f16 (at most 32 packed tokens, QK^T and PV on MFMA, one workgroup per head) against the plain fp32 torch
    reference of the same op: one text of 1..32 tokens, several short texts, the canonical padded shape (real texts + one-token dummies).
    Tolerance 3e-3: fp16 output, probabilities rounded to fp16 before the PV product."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    for heads, lens in ((16, [20]), (16, [1]), (8, [32]), (16, [16]), (16, [17]), (4, [5, 1, 9, 3]), (16, [22] + [1] * 10), (8, [7, 6, 1, 1, 1]), (2, [15, 16])):
        H, T = heads * 64, int(sum(lens))
        g = torch.Generator().manual_seed(100 + T + heads)
        qkv = (torch.randn((T, 3 * H), generator=g) * 1.5).half()
        first = np.cumsum(lens) - np.asarray(lens)
        tf = torch.from_numpy(np.repeat(first, lens).astype(np.int32)).cuda()
        qd = qkv.cuda()
        ctx = torch.full((T, H), float("nan"), dtype=torch.float16, device="cuda")
        rc = L.rdx_enc_attention_small_f16(0, qd.data_ptr(), tf.data_ptr(), T, heads, 64, 0.125, ctx.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, _lib.last_error()
        want = _attention_reference(qkv, [int(n) for n in lens], heads)
        torch.cuda.synchronize()
        got = ctx.float().cpu()
        assert torch.isfinite(got).all()
        assert float((got - want).abs().max()) <= 3e-3, (heads, lens, float((got - want).abs().max()))
    assert L.rdx_enc_attention_small_f16(0, qd.data_ptr(), tf.data_ptr(), 33, heads, 64, 0.125, ctx.data_ptr(), 0) != 0
    assert L.rdx_enc_attention_small_f16(0, qd.data_ptr(), tf.data_ptr(), T, heads, 32, 0.125, ctx.data_ptr(), 0) != 0


@pytest.mark.gpu
def test_encoder_attention_mfma_kernel_vs_fp32_reference():
    """rdx_enc_attention_mfma_f16 (flash-style attention of packed texts of any length: the corpus side, chunk texts of hundreds of
    tokens) against the plain fp32 torch reference of the same op: ragged lengths 1 .. 1500 on both sides of the 32-key tile and the
    64-query block, many short texts, one text only. Tolerance 3e-3: fp16 output, probabilities rounded to fp16 before PV."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(9)
    for heads, lens in ((16, [1, 64, 7, 200, 33, 2, 65, 31, 32, 128, 129]), (2, [1500, 1, 1023, 1024, 1025]), (4, list(rng.integers(1, 90, size=150))), (16, [300])):
        H, T = heads * 64, int(sum(lens))
        g = torch.Generator().manual_seed(T)
        qkv = (torch.randn((T, 3 * H), generator=g) * 1.5).half()
        lens_a = np.asarray(lens, dtype=np.int64)
        first = np.cumsum(lens_a) - lens_a
        nb = (lens_a + 63) // 64
        tix = np.repeat(np.arange(len(lens)), nb)
        q0 = (np.arange(int(nb.sum())) - np.repeat(np.cumsum(nb) - nb, nb)) * 64
        qb = torch.from_numpy(np.stack([first[tix], lens_a[tix], q0, np.zeros_like(q0)], axis=1).astype(np.int32)).cuda()
        qd = qkv.cuda()
        ctx = torch.full((T, H), float("nan"), dtype=torch.float16, device="cuda")
        rc = L.rdx_enc_attention_mfma_f16(0, qd.data_ptr(), qb.data_ptr(), int(qb.shape[0]), heads, 64, 0.125, ctx.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, _lib.last_error()
        want = _attention_reference(qkv, [int(n) for n in lens], heads)
        torch.cuda.synchronize()
        got = ctx.float().cpu()
        assert torch.isfinite(got).all()
        assert float((got - want).abs().max()) <= 3e-3, (heads, lens[:5], float((got - want).abs().max()))
    assert L.rdx_enc_attention_mfma_f16(0, qd.data_ptr(), qb.data_ptr(), 1, heads, 32, 0.125, ctx.data_ptr(), 0) != 0
    assert L.rdx_enc_attention_mfma_f16(0, qd.data_ptr(), qb.data_ptr(), 0, heads, 64, 0.125, ctx.data_ptr(), 0) == 0


@pytest.mark.gpu
def test_long_texts_take_the_packed_forward_with_the_mfma_attention():
    """chunk-like texts (up to several hundred tokens: what the indexer embeds, reference create_chromadb_index.py:300-387) stay on the
    packed forward — MFMA attention, add + LayerNorm kernels, no padding — and agree with the torch-operations forward and the module"""
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    rng = np.random.default_rng(11)
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(rng.choice(words, size=int(n))) for n in (400, 3, 70, 129, 64, 65, 250, 1, 33)]

    def make(fused, packed=True):
        p = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=512)
        p.fused_kernels, p.packed_forward = fused, packed
        return p.load()
    fast, plain, module = make(None), make(False), make(False, packed=False)
    assert fast._packed.long_attention
    got = fast.embed_device(texts)
    for ref in (plain.embed_device(texts), module.embed_device(texts)):
        cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=1)
        assert float((1 - cos).abs().max()) <= 1e-5, float((1 - cos).abs().max())
    assert fast.last_encode_stats["tokens_real"] == sum(len(t.split()) + 2 for t in texts)


@pytest.mark.gpu
def test_encoder_gelu_in_place_vs_the_framework_operation():
    """rdx_enc_gelu_f16 (E13: erf by Abramowitz & Stegun 7.1.26 in fp32, one rounding to fp16) against torch.nn.functional.gelu on fp16
    AND against the fp64 definition: random values across both branches, every fp16 bit pattern, infinities, NaN -> NaN. Tolerance: the
    neighbouring fp16 value (|diff| <= one fp16 ulp of the reference) or 1e-6 absolutely (the underflowing negative tail), whichever is
    larger. Of N(0, 9) inputs ~94 % come out as the framework's very fp16 value (the rest are its neighbour: the negative side, where
    gelu = x/2 erfc and 1.5e-7 of erf is a visible fraction of the result); the sign of a zero result is left open (the framework's own
    small- and large-tensor kernels disagree about gelu(-0.0))."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream

    def check(x):
        d = x.cuda()
        want = torch.nn.functional.gelu(d)
        got = d.clone()
        assert L.rdx_enc_gelu_f16(0, got.data_ptr(), got.numel(), st) == 0, _lib.last_error()
        torch.cuda.synchronize()
        nan = torch.isnan(want)
        assert bool((torch.isnan(got) == nan).all())
        w64, g64, x64 = want.double()[~nan], got.double()[~nan], d.double()[~nan]
        inf = torch.isinf(w64)
        assert bool((g64[inf] == w64[inf]).all())
        w64, g64, x64 = w64[~inf], g64[~inf], x64[~inf]
        assert bool(torch.isfinite(g64).all())
        exact = 0.5 * x64 * (1 + torch.erf(x64 * 2 ** -0.5))
        ulp = torch.clamp(2.0 ** (torch.floor(torch.log2(torch.clamp(w64.abs(), min=2.0 ** -14))) - 10), min=2.0 ** -24)
        tol = 1.000001 * torch.clamp(ulp, min=1e-6)
        assert bool(((g64 - w64).abs() <= tol).all()), float(((g64 - w64).abs() / tol).max())
        fin = torch.isfinite(exact) & (exact.abs() < 65504)
        assert bool(((g64 - exact).abs()[fin] <= tol[fin]).all())
        return float((g64 == w64).double().mean())
    for n in (8, 4096 * 33, 20_003 * 8):
        x = (torch.randn(n, generator=g) * 3).half()
        x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 6e-8, -10.0]).half()
        same = check(x)
        assert n == 8 or same >= 0.9, (n, same)
    allv = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).view(torch.float16)     # every fp16 bit pattern
    check(allv.repeat(64))                                          # (large enough for the framework's vectorised kernel)
    got = allv.cuda()
    assert L.rdx_enc_gelu_f16(0, got.data_ptr(), 12, st) != 0
    assert L.rdx_enc_gelu_f16(0, got.data_ptr() + 2, 8, st) != 0


@pytest.mark.gpu
def test_encoder_embed_and_last_layernorm_kernels():
    """rdx_enc_embed_f16 = the module's embedding sum (two fp16 adds in its order: bit-equal to torch); rdx_enc_layernorm_rows_f16 = torch's
    fp16 LayerNorm widened to fp32 (within one fp16 ulp of the fp32 LayerNorm)."""
    import torch
    from rag_dpo_amd import _lib
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    for H in (512, 1024, 2048):
        word = torch.randn((300, H), generator=g).half().cuda()
        pos = torch.randn((60, H), generator=g).half().cuda()
        typ = torch.randn((1, H), generator=g).half().cuda()
        tok = torch.randint(0, 300, (23,), generator=g).cuda()
        pid = torch.randint(0, 60, (23,), generator=g).cuda()
        out = torch.empty((23, H), dtype=torch.float16, device="cuda")
        assert L.rdx_enc_embed_f16(0, tok.data_ptr(), pid.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(), 23, H, out.data_ptr(), st) == 0, _lib.last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, word[tok] + pos[pid] + typ[0])
        s = (torch.randn((9, H), generator=g) * 2 + 0.5).half().cuda()
        gamma, beta = (torch.randn(H, generator=g) * 0.3 + 1).half().cuda(), (torch.randn(H, generator=g) * 0.2).half().cuda()
        o32 = torch.empty((9, H), dtype=torch.float32, device="cuda")
        assert L.rdx_enc_layernorm_rows_f16(0, s.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 9, H, o32.data_ptr(), st) == 0, _lib.last_error()
        torch.cuda.synchronize()
        ref = torch.nn.functional.layer_norm(s.float(), (H,), gamma.float(), beta.float(), 1e-5)
        assert bool(((o32 - ref).abs() <= 2.0 ** -10 * torch.clamp(ref.abs(), min=1.0)).all())
        assert torch.equal(o32, o32.half().float())               # values of an fp16 LayerNorm output
    assert L.rdx_enc_embed_f16(0, tok.data_ptr(), pid.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(), 23, 768, out.data_ptr(), 0) != 0


@pytest.mark.gpu
def test_single_question_forward_vs_module_forward():
    """the reference's online path, `embed_query` in front of every search (src/rag/retriever.py:150-154): the five-launches-per-layer
    forward of one question against `XLMRobertaModel.forward` of the same weights in fp16 AND in fp32 — |1 - cos| <= 1e-5 against the
    fp16 module, <= 2e-5 against fp32 — for questions on both sides of the 16-token block and for a question's sub-queries embedded
    together; the first call of a shape (eager) and its replays (HIP graph) return the same bits; full-size XLM-R-large once."""
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    words = [f"w{i}" for i in range(64)]
    qs = [" ".join(words[:n]) for n in (1, 5, 13, 14, 15, 22, 30)]
    for name in ("random-init:mid", "random-init:xlm-roberta-large"):
        fast = EmbeddingProvider(model_name=name, device="cuda:0", dtype=torch.float16).load()
        assert fast._packed.small_stage
        mod16 = EmbeddingProvider(model_name=name, device="cuda:0", dtype=torch.float16)
        mod16.packed_forward = False
        mod16.load()
        mod32 = None
        if name.endswith("mid"):
            mod32 = EmbeddingProvider(model_name=name, device="cuda:0", dtype=torch.float32)
            mod32.packed_forward = False
            mod32.load()
        for batch in [[q] for q in qs] + [[qs[1], qs[0], qs[2]], qs[:4]]:
            first = fast.embed_device(batch).clone()
            again = [fast.embed_device(batch).clone() for _ in range(3)]      # capture, replay, replay
            for a in again:
                assert torch.equal(a, first), batch
            ref = mod16.embed_device(batch)
            cos = torch.nn.functional.cosine_similarity(first.double(), ref.double(), dim=1)
            assert float((1 - cos).abs().max()) <= 1e-5, (name, batch, float((1 - cos).abs().max()))
            if mod32 is not None:
                cos = torch.nn.functional.cosine_similarity(first.double(), mod32.embed_device(batch).double(), dim=1)
                assert float((1 - cos).abs().max()) <= 2e-5, (name, batch, float((1 - cos).abs().max()))
        assert len(fast._packed._graph) >= 2
        for p in (fast, mod16, mod32):
            if p is not None:
                p.unload()


@pytest.mark.gpu
@pytest.mark.parametrize("mfma_min", [None, 1024])
def test_large_question_batches_replay_one_canonical_graph(mfma_min):
    """BASELINE config 5's encode leg: a large batch of questions is padded to the next multiple of 1024 tokens with one-token dummy
    texts and replays ONE captured graph per canonical shape; batches whose real token
    counts differ inside the same shape share it. Each batch against the eager forward of the same weights (|1 - cos| <= 1e-5)."""
    import torch
    from rag_dpo_amd import synth
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    fast = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=256).load()
    eager = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=256)
    eager.encoder_graphs = False
    eager.load()
    assert fast._packed.large_graphs and fast._packed.long_attention
    if mfma_min is not None:                                       # (developer knob: the MFMA attention kernel for question batches too —
        fast._packed.MFMA_MIN_TOKENS = mfma_min                    #  a work-unit list of fixed size rides in the canonical shape)
    seen_tokens = set()
    for seed in (1, 2, 3, 4, 5):
        texts = synth.query_texts(90, seed=seed)                   # ~20 tokens each: ~1.8 K tokens -> the 2048-token shape
        got = fast.embed_device(texts).clone()
        seen_tokens.add(fast.last_encode_stats["tokens_real"])
        assert 1024 < fast.last_encode_stats["tokens_real"] <= 2048
        ref = eager.embed_device(texts)
        cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=1)
        assert float((1 - cos).abs().max()) <= 1e-5, (seed, float((1 - cos).abs().max()))
    assert len(seen_tokens) >= 3                                   # different real token counts ...
    large = [k for k in fast._packed._graph if k[0] == "large"]
    assert len(large) == 1 and large[0][:3] == ("large", 90, 2048), large                        # ... one graph
    assert not eager._packed._graph
    fast.unload()
    eager.unload()


@pytest.mark.gpu
def test_unload_frees_the_model_memory():
    """`unload()` exists to give the VRAM back (reference src/utils/embedding_provider.py:107-114): modules, the packed encoder's
    concatenated QKV copies, captured graphs and their pools, scratch, pinned staging"""
    import gc
    import torch
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated(0)
    p = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16).load()
    for _ in range(3):
        p.embed(["w1 w2 w3"])
        p.embed(["w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11 w12 w13 w14 w15 w16 w17 w18", "w2"])
    p.embed([" ".join(f"w{i}" for i in range(90))] * 3)
    loaded = torch.cuda.memory_allocated(0)
    assert loaded > base + 10 * 2 ** 20 and len(p._packed._graph) >= 1
    p.unload()
    assert not p.is_loaded and p._packed is None and not p._pinned
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(0) <= base + 2 ** 20, (base, loaded, torch.cuda.memory_allocated(0))
    assert len(p.embed(["w1 w2"])[0]) == p.dims                  # and loads again on demand, like the reference's lazy load


@pytest.mark.gpu
def test_graph_capture_beside_a_search_on_another_thread(oracle):
    """the provider captures its small-batch graphs in thread-local capture mode: searches running on another thread of the process
    (one shared provider and collection serve concurrent sessions, reference app.py:42-43) allocate, launch and synchronise while a
    capture is open — neither side may fail, and the answers stay the oracle's"""
    import threading
    import torch
    from rag_dpo_amd import engine as eng, synth
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    corpus = synth.make_corpus(30_000, 256)
    q = synth.make_queries(8, 256, corpus)
    ix = eng.HipIndex(256)
    ix.add(corpus)
    es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q, 10, None)
    stop, errors, n_search = threading.Event(), [], [0]

    def searcher():
        try:
            while not stop.is_set():
                s, r, c = ix.search(q, 10)
                assert (r == er).all()
                n_search[0] += 1
        except Exception as e:                       # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=searcher)
    th.start()
    try:
        p = EmbeddingProvider(model_name="random-init:mid", device="cuda:0", dtype=torch.float16, batch_size=64).load()
        words = [f"w{i}" for i in range(40)]
        for n in range(1, 33):                        # many shapes: each is captured the second time it is seen
            texts = [" ".join(words[:n])] * (1 + n % 3)
            for _ in range(3):
                p.embed_device(texts)
        torch.cuda.synchronize()
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    assert p._packed.graphs == "auto" and len(p._packed._graph) >= 3 and n_search[0] > 0
    ix.close()
