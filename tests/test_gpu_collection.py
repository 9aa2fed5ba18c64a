"""The Chroma-shaped boundary on the real engine (librdx on cuda:0): same contract as the CPU run."""
import pytest

pytestmark = pytest.mark.gpu


def test_collection_contract_gpu():
    from test_collection import run_collection_contract
    col = run_collection_contract(None)          # None -> default factory = HipIndex
    from rag_dpo_amd.engine import HipIndex
    assert isinstance(col._engine, HipIndex)


def test_persistent_client_gpu(tmp_path):
    from rag_dpo_amd.collection import PersistentClient
    from test_collection import fill
    from rag_dpo_amd import synth
    cl = PersistentClient(path=str(tmp_path / "db"))
    col = cl.create_collection(name="rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb, ids, docs, metas = fill(col, n=300, dim=1024)
    q = synth.make_queries(2, 1024, emb).tolist()
    before = col.query(query_embeddings=q, n_results=50, where={"source": "CNIL"})
    cl.persist()
    col2 = PersistentClient(path=str(tmp_path / "db")).get_collection("rag_dpo_chunks")
    after = col2.query(query_embeddings=q, n_results=50, where={"source": "CNIL"})
    assert after["ids"] == before["ids"]
    assert after["distances"] == before["distances"]      # stored rows reloaded verbatim (rdx_index_add_stored): same floats
    import numpy as np
    assert (np.asarray(col2.get(include=["embeddings"])["embeddings"]) == np.asarray(col.get(include=["embeddings"])["embeddings"])).all()
    # what the scan has learned about this GPU's XCDs (speed only) travels with the store: set, persist, reopen, read back
    sh = [1.04, 0.97, 1.0, 1.01, 0.99, 1.02, 0.98, 0.99]
    got = col2._engine.xcd_shares(sh)
    assert abs(sum(got) - 8.0) < 1e-9 and all(abs(a - b) < 1e-9 for a, b in zip(got, sh))
    col2._write_header(col2._dir, col2._snap_rows)
    col3 = PersistentClient(path=str(tmp_path / "db")).get_collection("rag_dpo_chunks")
    assert all(abs(a - b) < 1e-4 for a, b in zip(col3._engine.xcd_shares(), sh))
    assert col3.query(query_embeddings=q, n_results=50, where={"source": "CNIL"})["ids"] == before["ids"]
    with pytest.raises(ValueError):
        col3._engine.xcd_shares([1, 1, 1, 1, 1, 1, 1, float("nan")])


def test_indexer_flow_gpu(tmp_path):
    """ingest -> verify -> reopen -> update on the HIP engine leaves the records (and the verify answers) the reference
    indexer left (tests/golden/indexer_golden.json)"""
    from test_indexer_golden import run_reset_then_update
    run_reset_then_update(tmp_path, None)


def test_indexer_device_embeddings_route(tmp_path):
    """ingest with device_embeddings=True (encoder output -> collection.add as a CUDA tensor) leaves the same ids, documents,
    metadata and neighbours as the reference-style list-of-lists route, survives a reopen through the journal, and the
    stored rows agree to one rounding of the second normalisation"""
    import numpy as np
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import indexer_world as IW
    from rag_dpo_amd import indexer as ix_mod
    from rag_dpo_amd.collection import PersistentClient
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    prov = EmbeddingProvider(model_name="random-init:tiny", device="cuda").load()
    chunks = [c for c in IW.base_chunks() if IW.POISON not in c.get("text", "")][:18]
    cols = []
    for route, dev in (("lists", False), ("device", True)):
        client = PersistentClient(path=str(tmp_path / route))
        ix = ix_mod.ChromaDBIndexer(client, prov, device_embeddings=dev)
        ix.init_chromadb("reset")
        ix.index_chunks(chunks, batch_size=5)
        assert ix.stats["chunks_indexed"] > 0
        cols.append(ix.collection)
    a = cols[0].get(include=["documents", "metadatas", "embeddings"])
    b = cols[1].get(include=["documents", "metadatas", "embeddings"])
    assert a["ids"] == b["ids"] and a["documents"] == b["documents"] and a["metadatas"] == b["metadatas"]
    np.testing.assert_allclose(np.asarray(a["embeddings"]), np.asarray(b["embeddings"]), rtol=0, atol=2.5e-7)
    q = prov.embed(["durée de conservation"])
    assert cols[0].query(query_embeddings=q, n_results=5)["ids"] == cols[1].query(query_embeddings=q, n_results=5)["ids"]
    again = PersistentClient(path=str(tmp_path / "device")).get_collection(ix_mod.COLLECTION_NAME)   # journal replay
    c = again.get(include=["embeddings"])
    assert c["ids"] == b["ids"]
    np.testing.assert_allclose(np.asarray(c["embeddings"]), np.asarray(b["embeddings"]), rtol=0, atol=2.5e-7)


def test_mask_cache_gpu():
    """the same filter/invalidations contract with HBM-resident rdx_mask objects; a stale mask is refused by the library"""
    import numpy as np
    from test_collection import run_mask_cache
    from rag_dpo_amd.engine import ResidentMask
    col = run_mask_cache(None)
    assert all(isinstance(r, ResidentMask) for _, r in col._mask_cache.values())
    eng = col._engine
    n = len(eng)
    m = eng.make_mask(np.full((n + 31) // 32, 0xFFFFFFFF, dtype=np.uint32))
    q = np.ones((1, eng.dim), dtype=np.float32)
    a = eng.search(q, 3, mask=m)
    b = eng.search(q, 3)
    assert (a[1] == b[1]).all() and (a[0] == b[0]).all()
    eng.add(np.ones((1, eng.dim), dtype=np.float32))
    import pytest
    with pytest.raises(Exception, match="mask"):
        eng.search(q, 3, mask=m)
