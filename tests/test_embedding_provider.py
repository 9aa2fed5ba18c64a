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
