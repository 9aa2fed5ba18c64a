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


def test_indexer_flow_gpu(tmp_path):
    """ingest -> verify -> reopen -> update on the HIP engine leaves the records (and the verify answers) the reference
    indexer left (tests/golden/indexer_golden.json)"""
    from test_indexer_golden import run_reset_then_update
    run_reset_then_update(tmp_path, None)
