"""EmbeddingProvider — same interface as reference src/utils/embedding_provider.py:34-191, MI355X backend.

Reference behaviour mirrored (file:line in /root/reference/src/utils/embedding_provider.py):
  constants DEFAULT_MODEL/DIMS/BATCH_SIZE/MAX_SEQ_LENGTH/TRUNCATE_CHARS            :25-31
  ctor kwargs model_name, device, dtype, batch_size, cache_dir; lazy model          :44-64
  dims / is_loaded properties, load() idempotent -> self, unload()                  :68-114
  embed(texts): [] -> []; char-truncate 20 000; encode batch; L2-normalise; tolist  :118-147
  embed_query, is_available, get_info, __repr__                                     :149-185

What differs underneath: the transformer forward is plain PyTorch-ROCm (plumbing: `transformers.XLMRobertaModel`,
CLS pooling as BGE-M3's dense head) and the L2-normalise is librdx K1 on the device (`rdx_l2_normalize`, the same
arithmetic the index uses for corpus rows). Weights and tokenizer are loaded ONLY from a local directory
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
            torch.manual_seed(0)
            model = XLMRobertaModel(XLMRobertaConfig(**cfg), add_pooling_layer=False)
            self._tokenizer = _HashTokenizer(cfg["vocab_size"], max_len=min(512, cfg["max_position_embeddings"] - 2))
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
        logger.info(f"{self.model_name} loaded in {time.time() - t0:.1f}s (dims={self._dims})")
        return self

    def unload(self):
        if self._model is not None:
            self._model = None
            self._tokenizer = None
            if str(self.device).startswith("cuda"):
                torch.cuda.empty_cache()

    @torch.no_grad()
    def _encode_raw(self, texts: List[str]) -> torch.Tensor:
        """un-normalised CLS embeddings, fp32, on the model's device, in input order"""
        order = sorted(range(len(texts)), key=lambda i: -len(texts[i]))   # length-sorted batches, like sentence-transformers
        out = torch.empty((len(texts), self._dims), dtype=torch.float32, device=self.device)
        for a in range(0, len(texts), self.batch_size):
            idx = order[a: a + self.batch_size]
            enc = self._tokenizer([texts[i] for i in idx])
            enc = {k: v.to(self.device) for k, v in enc.items()}
            hidden = self._model(**enc).last_hidden_state
            out[torch.tensor(idx, device=self.device)] = hidden[:, 0].to(torch.float32)   # CLS pooling (BGE-M3 dense)
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
