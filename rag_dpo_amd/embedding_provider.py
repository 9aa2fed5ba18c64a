"""EmbeddingProvider — same interface as reference src/utils/embedding_provider.py:34-191, MI355X backend.

Reference behaviour mirrored (file:line in /root/reference/src/utils/embedding_provider.py):
  constants DEFAULT_MODEL/DIMS/BATCH_SIZE/MAX_SEQ_LENGTH/TRUNCATE_CHARS            :25-31
  ctor kwargs model_name, device, dtype, batch_size, cache_dir; lazy model          :44-64
  dims / is_loaded properties, load() idempotent -> self, unload()                  :68-114
  embed(texts): [] -> []; char-truncate 20 000; encode batch; L2-normalise; tolist  :118-147
  embed_query, is_available, get_info, __repr__                                     :149-185

What differs underneath: the transformer forward runs over the checkpoint's `transformers.XLMRobertaModel` weights on the packed
real tokens of a batch (`_PackedEncoder`: GEMMs and GELU are PyTorch-ROCm plumbing; on a GPU in fp16 the attention and the
add + LayerNorm pairs are librdx kernels, `rdx_enc_attention_f16` / `rdx_enc_add_layernorm_f16`, for a single question the
projections too, `rdx_enc_linear_small_f16`; batches of up to 8 texts replay their forward as a HIP graph), CLS pooling as BGE-M3's dense
head, and the L2-normalise is librdx K1 on the device (`rdx_l2_normalize`, the same arithmetic the index uses for corpus rows). Weights and tokenizer are loaded ONLY from a local directory
(`model_name` itself, or `<cache_dir>/<model_name>` / HF-cache layout): this build never fetches by name
(no network; HF_HUB_OFFLINE). `model_name="random-init:xlm-roberta-large"` builds the BGE-M3 architecture with
random weights and a hashing tokenizer — shape/perf faithful for benchmarks, NOT value faithful
(encoder value parity is unpinned: no BGE-M3 weights exist offline).
"""
from __future__ import annotations

import logging
import os
import time
import zlib
from typing import List, Optional

import numpy as np
import torch

logger = logging.getLogger(__name__)

DEFAULT_MODEL = "BAAI/bge-m3"
DEFAULT_DIMS = 1024
DEFAULT_DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
DEFAULT_DTYPE = torch.float16 if torch.cuda.is_available() else torch.float32
DEFAULT_BATCH_SIZE = 64
MAX_SEQ_LENGTH = 8192
TRUNCATE_CHARS = 20000

# XLM-RoBERTa-large = BGE-M3's backbone (24 layers x 1024 hidden x 16 heads, FFN 4096, vocab 250 002)
_XLMR_LARGE = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                   intermediate_size=4096, max_position_embeddings=8194, type_vocab_size=1, pad_token_id=1,
                   bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5)


class _HashTokenizer:
    """whitespace pieces -> crc32 ids; only for random-init benchmarking (no sentencepiece model offline). The id of a piece is
    remembered (a real tokenizer's vocabulary lookup is a hash-table hit too) and the padded batch is assembled in numpy: 1024
    short questions take ~1.5 ms instead of 10 — the encode leg of BASELINE config 5 measures the GPU, not this stand-in."""

    def __init__(self, vocab_size: int, max_len: int = 512):
        self.vocab_size, self.max_len = vocab_size, max_len
        self._ids: dict = {}

    def __call__(self, texts: List[str]):
        cap, n = self.max_len - 2, len(texts)
        toks = [t.split()[:cap] for t in texts]
        lens = np.fromiter(map(len, toks), dtype=np.int64, count=n)
        words = [w for tk in toks for w in tk]
        ids = list(map(self._ids.get, words))              # vocabulary lookup at C speed; misses (None) are hashed once
        if None in ids:
            for j, v in enumerate(ids):
                if v is None:
                    w = words[j]
                    v = 4 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 4)
                    if len(self._ids) < 1_000_000:
                        self._ids[w] = v
                    ids[j] = v
        width = int(lens.max()) + 2 if n else 2
        inp = np.full((n, width), 1, dtype=np.int64)       # <pad> = 1
        if n:
            inp[:, 0] = 0                                  # <s>
            first = np.cumsum(lens) - lens
            row = np.repeat(np.arange(n), lens)
            inp[row, np.arange(len(words)) - np.repeat(first, lens) + 1] = np.asarray(ids, dtype=np.int64)
            inp[np.arange(n), lens + 1] = 2                # </s>
        att = (np.arange(width)[None, :] < (lens + 2)[:, None]).astype(np.int64)
        return {"input_ids": torch.from_numpy(inp), "attention_mask": torch.from_numpy(att)}


class _PackedEncoder:
    """The XLM-R forward over the REAL tokens only (PyTorch-ROCm plumbing, the checkpoint's own modules and weights).

    transformers pads a batch to its longest text and runs every token-wise operation — QKV / output / FFN projections, GELU,
    residual adds, LayerNorms: all but the attention itself — over the padding too (BASELINE config 5's 1024 questions: 28 672
    token slots for 20 649 tokens). Here the hidden states stay PACKED ([T_real][hidden]) through the whole stack; only around
    the attention are Q, K, V scattered into the padded [batch][seq] layout (index_copy) and the context gathered back
    (index_select). The three projections are ONE GEMM on concatenated weights. About half the launches of the module-by-module
    forward (the encode of a batch is launch-bound on a busy host) and 28 % fewer GEMM rows for that batch. Same arithmetic per
    token as `XLMRobertaModel.forward` (post-LayerNorm blocks, erf GELU, position ids = padding_idx + 1 + index in the text,
    attention over the text's own tokens only): tests/test_embedding_provider.py compares the two. With `fused` (fp16 on a GPU)
    the attention and the add + LayerNorm pairs are librdx kernels working on the packed layout directly: nothing is ever padded."""

    def __init__(self, model, fused: bool = False):
        e = model.embeddings
        self.word, self.pos, self.typ, self.ln, self.pad = e.word_embeddings, e.position_embeddings, e.token_type_embeddings, e.LayerNorm, int(e.padding_idx)
        cfg = model.config
        if getattr(cfg, "hidden_act", "gelu") != "gelu" or getattr(cfg, "position_embedding_type", None) not in (None, "absolute"):
            raise ValueError("packed forward: unsupported configuration")
        self.heads = int(cfg.num_attention_heads)
        self.hidden = int(cfg.hidden_size)
        self.layers = []
        for L in model.encoder.layer:
            a = L.attention
            wqkv = torch.cat([a.self.query.weight, a.self.key.weight, a.self.value.weight], 0).contiguous()
            bqkv = torch.cat([a.self.query.bias, a.self.key.bias, a.self.value.bias], 0).contiguous()
            self.layers.append((wqkv, bqkv, a.output.dense, a.output.LayerNorm, L.intermediate.dense, L.output.dense, L.output.LayerNorm))
        self._pad_buf: dict = {}
        self._graph: dict = {}
        self._seen: dict = {}
        # librdx's two encoder kernels (include/rdx.h: rdx_enc_attention_f16, rdx_enc_add_layernorm_f16) take the place of the
        # scatter -> padded attention -> transposing copy -> gather chain and of the add + LayerNorm pairs: fp16 on a GPU, 64-wide
        # heads, hidden a multiple of 512 up to 2048, texts up to FUSED_MAX_TOKENS tokens (the attention kernel is written for
        # questions: its work per token grows with the text). Anything else runs the torch operations below.
        self._lib = None
        self.fused = False
        self.small_linear = False
        self.small_stage = False
        p0 = self.layers[0][0]
        if fused and p0.is_cuda and p0.dtype == torch.float16 and self.hidden // self.heads == 64 and self.hidden % 512 == 0 and self.hidden <= 2048:
            from . import _lib
            self._lib = _lib.load()          # raises RdxUnavailable: a GPU provider asked for its kernels and the library is missing
            self._last_error = _lib.last_error
            self.fused = True
            inter0 = self.layers[0][4]
            self.small_linear = self.hidden % 512 == 0 and inter0.weight.shape[0] % 512 == 0   # (rdx_enc_linear_small_f16: inputs a multiple of 512 wide)
            # the five-launches-per-layer forward of one question (rdx_enc_stage_f16 & co., csrc/enc_small.hpp)
            self.small_stage = self.hidden in (512, 1024) and inter0.weight.shape[0] in (512, 1024, 2048, 4096)
            self.stage_fpb_o = int(os.environ.get("RDX_ENC_FPB_O", self.STAGE_FPB_O))
            self.stage_fpb_f2 = int(os.environ.get("RDX_ENC_FPB_F2", self.STAGE_FPB_F2))

    # Question batches (every text <= FUSED_MAX_TOKENS) of at least this many tokens would take the MFMA attention kernel too. Alone it
    # wins (1024 questions: 39 us per layer against the VALU kernel's 57, profiles/r04/attention_mfma_bench.txt); inside config 5's
    # pipeline it LOSES: the encode of batch i+1 runs beside the MFMA-bound search of batch i, and a kernel on the vector ALU fills what
    # the scan leaves idle while a second MFMA kernel queues for the same pipes (encode 16.1 / 16.0 against 15.4 / 15.4 ms, step 31.96 /
    # 31.76 against 31.15 / 31.24 ms, profiles/r04/c5_n1_bench.json, c5_mfma_attention_for_questions_n1_bench.json). Default: never; developer knob RDX_ENC_MFMA_MIN.
    MFMA_MIN_TOKENS = int(os.environ.get("RDX_ENC_MFMA_MIN", str(1 << 40)))
    FUSED_MAX_TOKENS = 64       # up to here the VALU attention kernel (written for questions); beyond, the MFMA kernel (long_attention)
    long_attention = os.environ.get("RDX_ENC_LONG_ATTN", "mfma") != "torch"   # developer: "torch" = scatter -> SDPA -> gather for texts beyond 64 tokens

    def _add_ln(self, a: torch.Tensor, b: torch.Tensor, ln) -> torch.Tensor:
        out = torch.empty_like(a)
        rc = self._lib.rdx_enc_add_layernorm_f16(a.device.index or 0, a.data_ptr(), b.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                                 float(ln.eps), a.shape[0], a.shape[1], out.data_ptr(),
                                                 torch.cuda.current_stream(a.device).cuda_stream)
        if rc:
            raise RuntimeError("rdx_enc_add_layernorm_f16: " + self._last_error())
        return out

    # The FFN's erf GELU is the framework's operation (bit-equal to the module forward). librdx's in-place kernel (E13, rdx_enc_gelu_f16: the
    # same values to within one fp16 ulp) is 81 against 101 us behind a 148 us FFN1 GEMM at 20 K tokens, and NOTHING in the pipeline: c5
    # encode 15.71 / 15.73 against 15.87 / 15.80 ms, ingest 3 429 against 3 435 chunks/s (profiles/r04/gelu_inplace_ab.txt) — opt-in only.
    inplace_gelu = os.environ.get("RDX_ENC_GELU", "torch") == "inplace"

    def _gelu(self, x: torch.Tensor) -> torch.Tensor:
        """erf GELU of the FFN's first projection: the framework's, or (RDX_ENC_GELU=inplace) librdx's in-place kernel E13"""
        if not self.inplace_gelu or x.numel() % 8 or not x.is_contiguous():
            return torch.nn.functional.gelu(x)
        if self._lib.rdx_enc_gelu_f16(x.device.index or 0, x.data_ptr(), x.numel(), torch.cuda.current_stream(x.device).cuda_stream):
            raise RuntimeError("rdx_enc_gelu_f16: " + self._last_error())
        return x

    # up to here the projections are librdx's weight-streaming kernel (rdx_enc_linear_small_f16) instead of the BLAS library's GEMM. Measured
    # (tools/enc_small_sweep.py, graph replay, XLM-R-large): one question (32 padded tokens) 1.71 -> 1.36 ms; at 64 tokens the two are
    # equal (1.74), beyond the kernel loses (every 16-feature workgroup re-reads all activations: 128 tokens 2.12 against 1.78 ms)
    SMALL_TOKENS = 32

    def _linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, gelu: bool = False) -> torch.Tensor:
        out = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
        rc = self._lib.rdx_enc_linear_small_f16(x.device.index or 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), x.shape[0], w.shape[0], w.shape[1],
                                                1 if gelu else 0, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        if rc:
            raise RuntimeError("rdx_enc_linear_small_f16: " + self._last_error())
        return out

    def _attention(self, qkv: torch.Tensor, tok_first: torch.Tensor, tok_len: torch.Tensor, max_len: int, qb=None) -> torch.Tensor:
        T = qkv.shape[0]
        ctx = torch.empty((T, self.hidden), dtype=qkv.dtype, device=qkv.device)
        if qb is not None:                # texts beyond FUSED_MAX_TOKENS (the corpus side): the flash-style MFMA kernel over 64-query blocks
            rc = self._lib.rdx_enc_attention_mfma_f16(qkv.device.index or 0, qkv.data_ptr(), qb.data_ptr(), int(qb.shape[0]), self.heads,
                                                      self.hidden // self.heads, (self.hidden // self.heads) ** -0.5, ctx.data_ptr(),
                                                      torch.cuda.current_stream(qkv.device).cuda_stream)
            if rc:
                raise RuntimeError("rdx_enc_attention_mfma_f16: " + self._last_error())
            return ctx
        rc = self._lib.rdx_enc_attention_f16(qkv.device.index or 0, qkv.data_ptr(), tok_first.data_ptr(), tok_len.data_ptr(), T, self.heads,
                                             self.hidden // self.heads, (self.hidden // self.heads) ** -0.5, int(max_len), ctx.data_ptr(),
                                             torch.cuda.current_stream(qkv.device).cuda_stream)
        if rc:
            raise RuntimeError("rdx_enc_attention_f16: " + self._last_error())
        return ctx

    # HIP-graph replay of the fused forward: captured per shape the second time the shape is seen, replayed with ONE launch; the five
    # small index tensors go into static device buffers first. graphs = "auto" (default): batches of at most SMALL_TEXTS texts — the
    # reference's online path: embed_query(), or the <= 4 sub-queries of one question embedded together (rag_dpo_amd/retriever.py) —
    # whose ~200 tiny kernels are pure launch latency (measured, one question on XLM-R-large fp16: module forward 6.3 ms, this
    # forward eager 3.4, replayed 1.55). Such a batch is padded to a CANONICAL shape so that the graphs are few and always hit: real
    # tokens up to a multiple of SMALL_TOKEN_GRANULE with one-token dummy texts (they attend to themselves; nobody reads their rows),
    # the CLS index list up to SMALL_TEXTS entries, the longest text up to 16 / 32 / 64 (it sizes the attention's LDS window): at
    # most 24 shapes. True: additionally every larger batch by its exact (texts, tokens, longest text) shape (a batch of 1024 gains
    # nothing on the GPU, 3 ms of host time; production batches rarely repeat a token count). False: never. At most MAX_GRAPHS shapes
    # are kept (least recently used out).
    graphs = "auto"
    large_graphs = os.environ.get("RDX_ENC_LARGE_GRAPHS", "1") != "0"   # canonical-shape graphs for large question batches too (cls())
    LARGE_TOKEN_GRANULE = 1024
    MAX_LARGE_GRAPHS = 3
    MAX_GRAPHS = 64
    SMALL_TEXTS = 8
    SMALL_TOKEN_GRANULE = 32

    # One question (at most STAGE_TOKENS packed tokens, padded to 16 or 32): five launches per layer, csrc/enc_small.hpp. The output
    # projection and FFN-down have 1024 features: with 16 per workgroup they would occupy 64 CUs, so their workgroups take 8 / 4 rows of
    # the MFMA tile (developer knobs RDX_ENC_FPB_O / RDX_ENC_FPB_F2; measured values in DESIGN.md §10).
    STAGE_TOKENS = 32
    STAGE_FPB_O = 8
    STAGE_FPB_F2 = 8

    def _stage(self, x, w, b, T, ln=None, y_out=None, res=None, rows=None, epi=0, fpb=0):
        N, K = int(w.shape[0]), int(w.shape[1])
        out = torch.empty((T, N), dtype=torch.float16, device=w.device)
        rc = self._lib.rdx_enc_stage_f16(w.device.index or 0, x.data_ptr(), rows.data_ptr() if rows is not None else None,
                                         ln.weight.data_ptr() if ln is not None else None, ln.bias.data_ptr() if ln is not None else None,
                                         float(ln.eps) if ln is not None else 0.0, y_out.data_ptr() if y_out is not None else None,
                                         w.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, T, N, K, epi, fpb,
                                         out.data_ptr(), torch.cuda.current_stream(w.device).cuda_stream)
        if rc:
            raise RuntimeError("rdx_enc_stage_f16: " + self._last_error())
        return out

    def _small_forward(self, tok, pos, first_d, tok_first) -> torch.Tensor:
        """[T <= 32] ids / positions -> fp32 [len(first_d)][hidden] CLS rows; librdx kernels only (no torch operation on the stream)"""
        lib, dev = self._lib, tok.device
        di, st = dev.index or 0, torch.cuda.current_stream(dev).cuda_stream
        T, H, n_cls = int(tok.shape[0]), self.hidden, int(first_d.shape[0])
        s = torch.empty((T, H), dtype=torch.float16, device=dev)
        if lib.rdx_enc_embed_f16(di, tok.data_ptr(), pos.data_ptr(), self.word.weight.data_ptr(), self.pos.weight.data_ptr(),
                                 self.typ.weight.data_ptr(), T, H, s.data_ptr(), st):
            raise RuntimeError("rdx_enc_embed_f16: " + self._last_error())
        ln, last = self.ln, len(self.layers) - 1
        for li, (wqkv, bqkv, dense_o, ln1, inter, out, ln2) in enumerate(self.layers):
            y = torch.empty((T, H), dtype=torch.float16, device=dev)
            qkv = self._stage(s, wqkv, bqkv, T, ln=ln, y_out=y, epi=0)
            ctx = torch.empty((T, H), dtype=torch.float16, device=dev)
            if lib.rdx_enc_attention_small_f16(di, qkv.data_ptr(), tok_first.data_ptr(), T, self.heads, H // self.heads,
                                               (H // self.heads) ** -0.5, ctx.data_ptr(), st):
                raise RuntimeError("rdx_enc_attention_small_f16: " + self._last_error())
            rows = None
            if li == last:                                   # everything behind the last attention is row-wise: only the CLS rows are needed
                rows, T = first_d, n_cls
            s1 = self._stage(ctx, dense_o.weight, dense_o.bias, T, res=y, rows=rows, epi=2, fpb=self.stage_fpb_o)
            y1 = torch.empty((T, H), dtype=torch.float16, device=dev)
            f = self._stage(s1, inter.weight, inter.bias, T, ln=ln1, y_out=y1, epi=1)
            s = self._stage(f, out.weight, out.bias, T, res=y1, epi=2, fpb=self.stage_fpb_f2)
            ln = ln2
        o = torch.empty((n_cls, H), dtype=torch.float32, device=dev)
        if lib.rdx_enc_layernorm_rows_f16(di, s.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps), n_cls, H, o.data_ptr(), st):
            raise RuntimeError("rdx_enc_layernorm_rows_f16: " + self._last_error())
        return o

    def _fused_forward(self, tok, pos, first_d, tok_first, tok_len, max_len: int = 0, qb=None) -> torch.Tensor:
        """the forward on packed tokens with librdx's kernels: [T] ids / positions -> fp32 [B][hidden] CLS rows"""
        if self.small_stage and tok.shape[0] <= self.STAGE_TOKENS:
            return self._small_forward(tok, pos, first_d, tok_first)
        F = torch.nn.functional
        x = self.ln(self.word(tok) + self.pos(pos) + self.typ.weight[0])                                         # [T][H]
        last = len(self.layers) - 1
        small = self.small_linear and x.shape[0] <= self.SMALL_TOKENS   # one question, a question's sub-queries: weight-streaming projections, GELU in the epilogue
        for li, (wqkv, bqkv, dense_o, ln1, inter, out, ln2) in enumerate(self.layers):
            qkv = self._linear(x, wqkv, bqkv) if small else F.linear(x, wqkv, bqkv)
            ctx = self._attention(qkv, tok_first, tok_len, max_len, qb)                                           # [T][H], no padding anywhere
            if li == last:                                   # everything behind the last attention is row-wise: only the CLS rows are needed
                ctx, x = ctx.index_select(0, first_d), x.index_select(0, first_d)
            if small:
                x = self._add_ln(self._linear(ctx, dense_o.weight, dense_o.bias), x, ln1)
                x = self._add_ln(self._linear(self._linear(x, inter.weight, inter.bias, gelu=True), out.weight, out.bias), x, ln2)
            else:
                x = self._add_ln(dense_o(ctx), x, ln1)
                x = self._add_ln(out(self._gelu(inter(x))), x, ln2)
        return x.to(torch.float32)

    def _replay(self, key, host: dict, to_dev, max_len: int, unpack=None):
        """-> the CLS rows from a captured graph of this shape, or None (shape not captured: the caller runs eagerly).
        unpack: the forward's index tensors as views of the one static buffer host["pk_blob"] is copied into"""
        args = (lambda st: unpack(st["pk_blob"])) if unpack is not None else (lambda st: tuple(st[n] for n in self._ORDER))
        ent = self._graph.pop(key, None)
        if ent is None:
            if len(self._seen) > 4096:
                self._seen.clear()
            self._seen[key] = self._seen.get(key, 0) + 1
            if self._seen[key] < 2:
                return None
            while len(self._graph) >= self.MAX_GRAPHS:
                torch.cuda.current_stream(self.layers[0][0].device).synchronize()   # (its last replay may still run: its pool is freed with it)
                self._graph.pop(next(iter(self._graph)))          # least recently used (dicts keep insertion order; a hit re-inserts)
            dev = self.layers[0][0].device
            static = {n: torch.empty(tuple(t.shape), dtype=t.dtype, device=dev) for n, t in host.items()}
            for n, t in host.items():
                to_dev(n, t, static[n])
            try:
                side = torch.cuda.Stream(device=dev)             # one eager run on a side stream first (library workspaces), as torch asks
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._fused_forward(*args(static), max_len, static.get("pk_qb"))
                torch.cuda.current_stream(dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: only THIS thread's calls are restricted while the capture runs — a search another
                # thread has in flight on the same device (one shared provider and collection serve concurrent sessions, reference
                # app.py:42-43) may allocate and synchronise as it likes
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = self._fused_forward(*args(static), max_len, static.get("pk_qb"))
            except Exception as e:                               # noqa: BLE001  (a capture that fails costs speed only: eager from now on)
                logger.warning(f"encoder graph capture failed ({e!r}); the forward stays eager")
                self.graphs = False
                return None
            ent = (g, static, out)
        self._graph[key] = ent
        g, static, out = ent
        for n, t in host.items():
            to_dev(n, t, static[n])
        g.replay()
        return out                                               # (overwritten by the next replay of this shape: consume it on the stream)

    _ORDER = ("pk_tok", "pk_pos", "pk_first", "pk_tfirst", "pk_tlen")

    @staticmethod
    def _query_blocks(first: np.ndarray, lens: np.ndarray) -> torch.Tensor:
        """The MFMA attention kernel's work units: one per 64 queries of a text, {first token, length, first query, 0}."""
        nb = (lens + 63) // 64
        tix = np.repeat(np.arange(len(lens), dtype=np.int64), nb)
        q0 = (np.arange(int(nb.sum()), dtype=np.int64) - np.repeat(np.cumsum(nb) - nb, nb)) * 64
        return torch.from_numpy(np.stack([first[tix], lens[tix], q0, np.zeros_like(q0)], axis=1).astype(np.int32))

    @torch.no_grad()
    def cls(self, ids: torch.Tensor, lens: np.ndarray, to_dev) -> torch.Tensor:
        """ids: [B][S] int64 on the host, right-padded; lens[b] = tokens of text b (>= 1). -> fp32 [B][hidden] CLS rows on the device.
        to_dev(name, host tensor[, out]) -> device tensor (the provider's pinned, non-blocking copies)."""
        F = torch.nn.functional
        B, S = int(ids.shape[0]), int(ids.shape[1])
        lens = np.asarray(lens, dtype=np.int64)
        T = int(lens.sum())
        first = np.cumsum(lens) - lens                                   # packed index of every text's first token (CLS)
        row = np.repeat(np.arange(B, dtype=np.int64), lens)
        col = np.arange(T, dtype=np.int64) - np.repeat(first, lens)
        ids_np = ids.numpy()
        host = {"pk_tok": torch.from_numpy(np.ascontiguousarray(ids_np[row, col])), "pk_pos": torch.from_numpy(col + (self.pad + 1)),
                "pk_first": torch.from_numpy(first)}
        H, nh = self.hidden, self.heads
        if self.fused and (int(lens.max()) <= self.FUSED_MAX_TOKENS or self.long_attention):
            host["pk_tfirst"] = torch.from_numpy(np.repeat(first, lens).astype(np.int32))
            host["pk_tlen"] = torch.from_numpy(np.repeat(lens, lens).astype(np.int32))
            max_len = int(lens.max())
            if max_len > self.FUSED_MAX_TOKENS or (self.long_attention and T >= self.MFMA_MIN_TOKENS):
                # the corpus side (chunk texts of hundreds of tokens; and large question batches when MFMA_MIN_TOKENS says so): the MFMA
                # kernel, one work unit per 64 queries
                host["pk_qb"] = self._query_blocks(first, lens)
            if self.graphs and B <= self.SMALL_TEXTS and max_len <= self.FUSED_MAX_TOKENS:
                # (<= 32 tokens run the stage kernels: their cost follows the activation rows a workgroup stages, so the canonical shapes
                #  are 8, 16, 24 and 32 tokens — a typical 20-token question pays for 24 rows, not 32)
                g = 8 if (self.small_stage and T <= self.STAGE_TOKENS) else self.SMALL_TOKEN_GRANULE
                Tp = -(-T // g) * g                               # one-token dummy texts behind the real ones
                lb = 16 if max_len <= 16 else (32 if max_len <= 32 else 64)
                # the five index arrays of the canonical shape in ONE buffer: one pinned copy per question instead of five
                # (each small copy is ~15 us of stream time: 0.07 of a 0.95 ms embed_query)
                o1, o2, o3, o4, nb_ = 8 * Tp, 16 * Tp, 16 * Tp + 8 * self.SMALL_TEXTS, 20 * Tp + 8 * self.SMALL_TEXTS, 24 * Tp + 8 * self.SMALL_TEXTS
                blob = np.empty(nb_, dtype=np.uint8)
                v64, v32 = blob[:o3].view(np.int64), blob[o3:].view(np.int32)
                v64[:T] = ids_np[row, col]
                v64[T:Tp] = self.pad
                v64[Tp:Tp + T] = col + (self.pad + 1)
                v64[Tp + T:2 * Tp] = self.pad + 1
                v64[2 * Tp:2 * Tp + B] = first
                v64[2 * Tp + B:] = 0
                v32[:T] = np.repeat(first, lens)
                v32[T:Tp] = np.arange(T, Tp)
                v32[Tp:Tp + T] = np.repeat(lens, lens)
                v32[Tp + T:] = 1

                def unpack(d):
                    return (d[:o1].view(torch.int64), d[o1:o2].view(torch.int64), d[o2:o3].view(torch.int64), d[o3:o4].view(torch.int32),
                            d[o4:].view(torch.int32))
                hb = {"pk_blob": torch.from_numpy(blob)}
                out = self._replay(("small", Tp, lb), hb, to_dev, lb, unpack)
                if out is None:   # shape not captured yet: the SAME padded tensors eagerly, so that call 1 and the replays run identical shapes
                    out = self._fused_forward(*unpack(to_dev("pk_blob", hb["pk_blob"])), lb)
                return out[:B]
            elif self.graphs and self.large_graphs and max_len <= self.FUSED_MAX_TOKENS and T >= self.LARGE_TOKEN_GRANULE:
                # a large batch of questions (BASELINE config 5: 1024 texts, ~20 K tokens): its ~230 launches cost a busy host 15 - 35 ms
                # per encode (measured) against 15 ms of GPU time. Canonical shape = real tokens padded to a multiple of
                # LARGE_TOKEN_GRANULE with one-token dummy texts (<= 5 % more rows at 20 K tokens; nobody reads their outputs), the
                # longest text rounded to 16 / 32 / 64: consecutive batches of a serving loop hit the same graph, ONE launch per encode.
                g = self.LARGE_TOKEN_GRANULE
                Tp = -(-T // g) * g
                extra = Tp - T
                lb = 16 if max_len <= 16 else (32 if max_len <= 32 else 64)
                padded = {"pk_tok": torch.from_numpy(np.concatenate([host["pk_tok"].numpy(), np.full(extra, self.pad, dtype=np.int64)])),
                          "pk_pos": torch.from_numpy(np.concatenate([host["pk_pos"].numpy(), np.full(extra, self.pad + 1, dtype=np.int64)])),
                          "pk_first": host["pk_first"],
                          "pk_tfirst": torch.from_numpy(np.concatenate([host["pk_tfirst"].numpy(), np.arange(T, Tp, dtype=np.int32)])),
                          "pk_tlen": torch.from_numpy(np.concatenate([host["pk_tlen"].numpy(), np.ones(extra, dtype=np.int32)]))}
                if "pk_qb" in host:
                    # (B + extra work units; `extra` moves with T inside one canonical shape, so the list is filled up to B + g units
                    #  with repeats of the last one: the same rows written twice with the same values)
                    qb = self._query_blocks(np.concatenate([first, np.arange(T, Tp, dtype=np.int64)]),
                                            np.concatenate([lens, np.ones(extra, dtype=np.int64)]))
                    padded["pk_qb"] = torch.cat([qb, qb[-1:].expand(B + g - qb.shape[0], 4)]).contiguous()
                big = [k_ for k_ in self._graph if k_[0] == "large"]
                key = ("large", B, Tp, lb)
                if key not in self._graph and len(big) >= self.MAX_LARGE_GRAPHS:
                    torch.cuda.current_stream(self.layers[0][0].device).synchronize()   # (its last replay may still run)
                    self._graph.pop(big[0])                       # each holds the activations of ~Tp tokens: keep few
                out = self._replay(key, padded, to_dev, lb)
                if out is not None:
                    return out
                return self._fused_forward(*(to_dev(n, padded[n]) for n in self._ORDER), lb, to_dev("pk_qb", padded["pk_qb"]) if "pk_qb" in padded else None)
            elif self.graphs is True:
                nqb = int(host["pk_qb"].shape[0]) if "pk_qb" in host else 0
                out = self._replay((B, T, max_len, nqb), host, to_dev, max_len)   # (the longest text sizes the attention's LDS window: part of the shape)
                if out is not None:
                    return out
            return self._fused_forward(*(to_dev(n, host[n]) for n in self._ORDER), max_len, to_dev("pk_qb", host["pk_qb"]) if "pk_qb" in host else None)
        tok, pos, first_d = (to_dev(n, host[n]) for n in ("pk_tok", "pk_pos", "pk_first"))
        x = self.ln(self.word(tok) + self.pos(pos) + self.typ.weight[0])                                         # [T][H]
        flat_d = to_dev("pk_flat", torch.from_numpy(row * S + col))                                              # slot of packed token t in the padded [B*S] layout
        kmask = to_dev("pk_mask", torch.from_numpy(np.arange(S)[None, :] < lens[:, None])).view(B, 1, 1, S)       # keys of the text itself
        key = (B, S, x.dtype, x.device)
        qkv_pad = self._pad_buf.get(key)
        if qkv_pad is None:
            if len(self._pad_buf) > 8:
                self._pad_buf.clear()
            qkv_pad = self._pad_buf[key] = torch.zeros((B * S, 3 * H), dtype=x.dtype, device=x.device)   # (stale padding slots are masked keys / dropped queries)
        last = len(self.layers) - 1
        for li, (wqkv, bqkv, dense_o, ln1, inter, out, ln2) in enumerate(self.layers):
            qkv_pad.index_copy_(0, flat_d, F.linear(x, wqkv, bqkv))
            q, k, v = qkv_pad.view(B, S, 3, nh, H // nh).permute(2, 0, 3, 1, 4)                                   # [B][heads][S][head_dim] views
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=kmask)
            ctx = ctx.transpose(1, 2).reshape(B * S, H).index_select(0, flat_d)                                   # back to [T][H]
            if li == last:                                   # everything behind the last attention is row-wise: only the CLS rows are needed
                ctx, x = ctx.index_select(0, first_d), x.index_select(0, first_d)
            x = ln1(dense_o(ctx) + x)
            x = ln2(out(F.gelu(inter(x))) + x)
        return x.to(torch.float32)


def _resolve_local_dir(model_name: str, cache_dir: Optional[str]) -> Optional[str]:
    cands = [model_name]
    if cache_dir:
        cands += [os.path.join(cache_dir, model_name), os.path.join(cache_dir, model_name.replace("/", "_")),
                  os.path.join(cache_dir, "models--" + model_name.replace("/", "--"))]
    for c in cands:
        if os.path.isdir(c):
            if os.path.exists(os.path.join(c, "config.json")):
                return c
            snaps = os.path.join(c, "snapshots")   # HF hub cache layout
            if os.path.isdir(snaps):
                for s in sorted(os.listdir(snaps)):
                    if os.path.exists(os.path.join(snaps, s, "config.json")):
                        return os.path.join(snaps, s)
    return None


class EmbeddingProvider:
    """Dense BGE-M3 embeddings, L2-normalised (unit rows), d = 1024. Calls are synchronous and thread-safe."""

    def __init__(self, model_name: str = DEFAULT_MODEL, device: str = DEFAULT_DEVICE, dtype: torch.dtype = DEFAULT_DTYPE,
                 batch_size: int = DEFAULT_BATCH_SIZE, cache_dir: Optional[str] = None):
        self.model_name = model_name
        self.device = device
        self.dtype = dtype
        self.batch_size = batch_size
        self.cache_dir = cache_dir
        self._model = None
        self._tokenizer = None
        self._dims: int = DEFAULT_DIMS
        import threading
        self._lock = threading.Lock()
        self._pinned: dict = {}
        self._packed = None
        logger.info(f"EmbeddingProvider configured: {model_name} ({device}, {dtype}, batch={batch_size})")

    @property
    def dims(self) -> int:
        return self._dims

    @property
    def is_loaded(self) -> bool:
        return self._model is not None

    def load(self) -> "EmbeddingProvider":
        if self._model is not None:
            return self
        t0 = time.time()
        from transformers import XLMRobertaConfig, XLMRobertaModel
        if self.model_name.startswith("random-init:"):
            spec = self.model_name.split(":", 1)[1]
            cfg = dict(_XLMR_LARGE)
            if spec.startswith("tiny"):     # tests: same architecture, toy size
                cfg.update(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=1000,
                           max_position_embeddings=514)
            elif spec.startswith("mid"):    # tests of the fused kernels: 64-wide heads, hidden a multiple of 512
                cfg.update(hidden_size=512, num_hidden_layers=3, num_attention_heads=8, intermediate_size=1024, vocab_size=1000,
                           max_position_embeddings=514)
            torch.manual_seed(0)
            model = XLMRobertaModel(XLMRobertaConfig(**cfg), add_pooling_layer=False)
            self._tokenizer = _HashTokenizer(cfg["vocab_size"], max_len=min(MAX_SEQ_LENGTH, cfg["max_position_embeddings"] - 2))   # (the reference caps texts at 8192 tokens, embedding_provider.py:30)
        else:
            local = _resolve_local_dir(self.model_name, self.cache_dir)
            if local is None:
                raise RuntimeError(
                    f"EmbeddingProvider: no local checkpoint for '{self.model_name}' (looked in the name itself and under "
                    f"cache_dir={self.cache_dir!r}). This build never downloads models; pass a directory holding "
                    "config.json + weights + tokenizer files of BAAI/bge-m3.")
            from transformers import AutoTokenizer
            model = XLMRobertaModel.from_pretrained(local, add_pooling_layer=False, local_files_only=True)
            tok = AutoTokenizer.from_pretrained(local, local_files_only=True)
            self._tokenizer = lambda texts: tok(texts, padding=True, truncation=True, max_length=MAX_SEQ_LENGTH, return_tensors="pt")
        self._model = model.to(device=self.device, dtype=self.dtype).eval()
        self._dims = int(self._model.config.hidden_size)
        self._packed = None
        if self.packed_forward:
            try:
                fused = self.fused_kernels if self.fused_kernels is not None else (str(self.device).startswith("cuda") and self.dtype == torch.float16)
                self._packed = _PackedEncoder(self._model, fused=bool(fused))
                self._packed.graphs = "auto" if self.encoder_graphs is None else bool(self.encoder_graphs)
            except ValueError as e:                                           # another architecture: the module forward stays
                logger.info(f"packed forward not available for this model ({e}); using the module forward")
        logger.info(f"{self.model_name} loaded in {time.time() - t0:.1f}s (dims={self._dims})")
        return self

    def unload(self):
        """frees the model's VRAM (reference src/utils/embedding_provider.py:107-114): the modules, the packed encoder's
        concatenated QKV weights, its captured graphs (each holds a private memory pool) and scratch, the pinned staging rings"""
        with self._lock:
            if self._model is not None or self._packed is not None:
                if self._packed is not None:
                    self._packed._graph.clear()       # graphs first: their pools go back to the allocator
                    self._packed._pad_buf.clear()
                    self._packed.layers = []
                self._packed = None
                self._pinned.clear()
                self._model = None
                self._tokenizer = None
                if str(self.device).startswith("cuda"):
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()

    # Length buckets. sentence-transformers (the reference's encoder, src/utils/embedding_provider.py:139-145) sorts a call's texts by
    # length and pads every batch of `batch_size` to ITS longest text. With the large batches a GPU wants (BASELINE config 5 hands
    # 1024 query texts to one call) one batch means one width: 8-24-word questions padded to the longest are ~30 % padding
    # tokens. A batch is therefore cut, after tokenising, into at most `max_buckets` buckets of consecutive (token-count-sorted)
    # rows, each forwarded at its own width; the cuts (multiples of 64 rows) minimise padded tokens + a per-forward charge.
    encoder_graphs: Optional[bool] = None  # HIP-graph replay of the fused forward: None = batches of up to 8 texts (embed_query, a question's sub-queries), True = every batch, False = never (_PackedEncoder.graphs)
    fused_kernels: Optional[bool] = None   # None: librdx's encoder kernels whenever the provider runs fp16 on a GPU (the library must load); False: torch operations only
    packed_forward = True              # _PackedEncoder: token-wise layers over the real tokens only (padding only around the attention)
    max_buckets = 4
    bucket_granule = 64
    bucket_overhead_tokens = 4096      # what one more forward costs, in token-equivalents. Measured on MI355X, XLM-R-large fp16, 1024 questions:
                                       # ONE forward of 1024 x 28 tokens 23.5 ms; TWO of 512 x 28 + 512 x 20 (14 % fewer tokens) 13.1 + 10.4 = 23.5 ms —
                                       # the smaller GEMMs and the second pass of ~400 launches eat what the padding saved: ~4 K tokens per forward
    time_buckets = False               # True: CUDA events around every bucket's forward (last_encode_stats["buckets"][i]["ms"])
    last_encode_stats: Optional[dict] = None

    def _h2d(self, name: str, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """small host tensor -> device WITHOUT blocking the host: through pinned staging buffers (a ring of PIN_RING per name and
        shape, each guarded by an event: it is not overwritten before the copy that read it last has run). A pageable
        `.to(device)` is a synchronous copy: it parks the host until everything enqueued on the stream so far — the previous batch's
        search — has finished, and the launches of the forward then start late (BASELINE config 5's pipeline: the encode of batch
        i+1 is issued behind search i, while the GPU may still be running the encode of batch i). A ring, not one buffer: the copy a
        buffer waits for belongs to the encode before the previous one — with a single buffer the "rows" copy at the END of encode i
        made the host wait for the whole of encode i before it could issue encode i+1 (measured: 12 of 17 ms of host time per step)."""
        if not str(self.device).startswith("cuda"):
            return t.to(self.device) if out is None else out.copy_(t)
        key = (name, tuple(t.shape), t.dtype)
        ent = self._pinned.get(key)
        if ent is None:
            if len(self._pinned) > 64:
                self._pinned.clear()
            ent = self._pinned[key] = [0, [[torch.empty(t.shape, dtype=t.dtype).pin_memory(), None] for _ in range(self.PIN_RING)]]
        slot = ent[1][ent[0]]
        ent[0] = (ent[0] + 1) % self.PIN_RING
        buf, ev = slot
        if ev is not None:
            ev.synchronize()
        buf.copy_(t)
        d = buf.to(self.device, non_blocking=True) if out is None else out.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return d

    PIN_RING = 3

    def _forward_cls(self, feed: dict) -> torch.Tensor:
        """fp32 CLS rows of one padded bucket. (Round 3 also built a HIP-graph replay of this forward — ~700 launches that a busy host
        issues more slowly (16 - 57 ms measured on this pool's boxes) than the GPU runs them (22 ms): captured per (rows, width)
        shape, replayed with one launch. Alone it ran as fast as eager (27.8 ms per embed_device); enqueued behind a running search,
        as BASELINE config 5's pipeline does, the replayed forward took 49 ms instead of 24. Removed: eager it stays.)"""
        return self._model(**feed).last_hidden_state[:, 0].to(torch.float32)

    def _bucket_cuts(self, lens_desc: np.ndarray) -> List[int]:
        """row boundaries [0, ..., n] of the buckets for token counts sorted in descending order"""
        n = int(lens_desc.shape[0])
        g = max(1, int(self.bucket_granule))
        cand = list(range(0, n, g)) + [n]
        m, B = len(cand) - 1, max(1, int(self.max_buckets))
        if m <= 1 or B == 1:
            return [0, n]
        INF = float("inf")
        cost = [[INF] * (m + 1) for _ in range(B + 1)]    # cost[b][j]: rows [0, cand[j]) in exactly b buckets
        prev = [[-1] * (m + 1) for _ in range(B + 1)]
        cost[0][0] = 0.0
        for b in range(1, B + 1):
            for j in range(1, m + 1):
                for i in range(b - 1, j):
                    if cost[b - 1][i] == INF:
                        continue
                    c = cost[b - 1][i] + (cand[j] - cand[i]) * float(lens_desc[cand[i]]) + self.bucket_overhead_tokens
                    if c < cost[b][j]:
                        cost[b][j], prev[b][j] = c, i
        b = min(range(1, B + 1), key=lambda x: cost[x][m])
        cuts, j = [n], m
        while b > 0:
            j = prev[b][j]
            cuts.append(cand[j])
            b -= 1
        return cuts[::-1]

    @torch.no_grad()
    def _encode_raw(self, texts: List[str]) -> torch.Tensor:
        """un-normalised CLS embeddings, fp32, on the model's device, in input order"""
        if self._packed is not None and len(texts) <= self._packed.SMALL_TEXTS and str(self.device).startswith("cuda"):
            # the online path — one question, or a question's sub-queries: nothing to sort (the packed forward pads nothing), no row
            # permutation to undo: tokenise, ONE pinned copy, the replayed forward, one device copy of the rows handed back
            enc = self._tokenizer(list(texts))
            lens_np = enc["attention_mask"].numpy().sum(axis=1)   # (numpy, not torch: see _encode_raw's note on host-side tensor ops)
            cls = self._packed.cls(enc["input_ids"], lens_np, self._h2d)
            self.last_encode_stats = {"texts": len(texts), "tokens_real": int(lens_np.sum()), "tokens_padded": int(lens_np.sum()),
                                      "tokens_padded_one_width": int(enc["input_ids"].numel()), "buckets": [{"rows": len(texts), "width": int(lens_np.max())}],
                                      "real_over_padded": 1.0}
            return cls.clone()                                    # (the forward's output buffer belongs to its graph: the next replay overwrites it)
        order = sorted(range(len(texts)), key=lambda i: -len(texts[i]))   # length-sorted batches, like sentence-transformers
        out = torch.empty((len(texts), self._dims), dtype=torch.float32, device=self.device)
        stats = {"texts": len(texts), "tokens_real": 0, "tokens_padded": 0, "tokens_padded_one_width": 0, "buckets": []}
        events = []
        on_gpu = str(self.device).startswith("cuda")
        for a in range(0, len(texts), self.batch_size):
            idx = order[a: a + self.batch_size]
            enc = self._tokenizer([texts[i] for i in idx])
            ids, att = enc["input_ids"], enc["attention_mask"]
            # Host-side bookkeeping in NUMPY. torch's CPU operations run on its intra-op thread pool, and on a box whose cgroup grants
            # fewer cores than the pool has threads a reduction over a [64][1000] mask takes 50 ms instead of 30 us (measured: this
            # container, 8 threads; `attention_mask.sum(dim=1)` alone). That — not launches — was the "busy host" of DESIGN.md §10's
            # c5 numbers (an encode issued in 15 - 35 ms) and 2/3 of an ingest batch's wall time.
            lens_all = att.numpy().sum(axis=1)
            by_np = np.argsort(-lens_all, kind="stable")                     # characters were a proxy: now by token count
            by_len = torch.from_numpy(by_np)
            ids, att, lens_np = torch.from_numpy(ids.numpy()[by_np]), torch.from_numpy(att.numpy()[by_np]), lens_all[by_np]
            rows = [idx[i] for i in by_np.tolist()]
            cuts = [0, int(lens_np.shape[0])] if self._packed is not None else self._bucket_cuts(lens_np)   # (packed: padding costs the attention only)
            stats["tokens_real"] += int(lens_np.sum())
            stats["tokens_padded_one_width"] += int(ids.shape[0] * ids.shape[1])
            extra = {k: v[by_len] for k, v in enc.items() if k not in ("input_ids", "attention_mask")}
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                w = int(lens_np[lo])                                        # the bucket's longest row (right padding: columns [:w])
                feed = None
                if self._packed is None:
                    feed = {"input_ids": self._h2d("ids", ids[lo:hi, :w]), "attention_mask": self._h2d("att", att[lo:hi, :w])}
                    feed.update({k: self._h2d(k, v[lo:hi, :w]) for k, v in extra.items()})
                if self.time_buckets and on_gpu:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if self._packed is not None:                                                     # CLS pooling (BGE-M3 dense)
                    cls = self._packed.cls(ids[lo:hi, :w], lens_np[lo:hi], self._h2d)
                else:
                    cls = self._forward_cls(feed)
                out[self._h2d("rows", torch.tensor(rows[lo:hi], dtype=torch.int64))] = cls
                if self.time_buckets and on_gpu:
                    e1.record()
                    events.append((len(stats["buckets"]), e0, e1))
                stats["tokens_padded"] += (hi - lo) * w
                stats["buckets"].append({"rows": hi - lo, "width": w})
        if events:
            torch.cuda.synchronize()
            for i, e0, e1 in events:
                stats["buckets"][i]["ms"] = round(e0.elapsed_time(e1), 3)
        stats["real_over_padded"] = round(stats["tokens_real"] / max(1, stats["tokens_padded"]), 4)
        self.last_encode_stats = stats
        return out

    def embed(self, texts: List[str]) -> List[List[float]]:
        if not texts:
            return []
        with self._lock:
            if self._model is None:
                self.load()
            truncated = [t[:TRUNCATE_CHARS] if len(t) > TRUNCATE_CHARS else t for t in texts]
            raw = self._encode_raw(truncated)
            return self._normalize(raw).tolist()

    def embed_device(self, texts: List[str]) -> torch.Tensor:
        """the same encode as embed(), left on the device: [T, dims] fp32 CLS embeddings, NOT yet normalised. For callers
        that hand the batch straight to the device index (HipIndex.add / search_device run K1 on it anyway), which
        skips embed()'s T x 1024 Python-float round trip."""
        if not texts:
            return torch.empty((0, self._dims), dtype=torch.float32, device=self.device)
        with self._lock:
            if self._model is None:
                self.load()
            return self._encode_raw([t[:TRUNCATE_CHARS] if len(t) > TRUNCATE_CHARS else t for t in texts])

    def _normalize(self, raw: torch.Tensor) -> np.ndarray:
        """K1 on the device (librdx); x / max(|x|, 1e-12), the arithmetic the index uses for corpus rows"""
        from . import _lib as L
        import ctypes
        if not raw.is_cuda:
            raise L.RdxUnavailable("EmbeddingProvider needs the model on an MI355X (device='cuda'): the L2-normalise step "
                                   "runs in librdx; there is no CPU path")
        lib = L.load(require_gpu=True)
        raw = raw.contiguous()
        out = torch.empty_like(raw)
        stream = torch.cuda.current_stream(raw.device).cuda_stream
        L.check(lib.rdx_l2_normalize(raw.device.index or 0, ctypes.c_void_p(raw.data_ptr()), raw.shape[0], raw.shape[1],
                                     ctypes.c_void_p(out.data_ptr()), L.RDX_DEVICE, ctypes.c_void_p(stream)))
        return out.cpu().numpy()

    def embed_query(self, query: str) -> List[float]:
        return self.embed([query])[0]

    def is_available(self) -> bool:
        try:
            if str(self.device).startswith("cuda") and not torch.cuda.is_available():
                return False
            return True
        except Exception:
            return False

    def get_info(self) -> dict:
        vram_gb = torch.cuda.memory_allocated(0) / 1024 ** 3 if str(self.device).startswith("cuda") and torch.cuda.is_available() else 0
        return {"model": self.model_name, "device": self.device, "dtype": str(self.dtype), "dims": self._dims,
                "loaded": self.is_loaded, "vram_gb": round(vram_gb, 2), "batch_size": self.batch_size}

    def __repr__(self) -> str:
        return f"EmbeddingProvider({self.model_name}, {self.device}, {'loaded' if self.is_loaded else 'not loaded'})"

    def __del__(self):
        try:
            self.unload()
        except Exception:
            pass
