"""Host logic of the Chroma-shaped boundary (ids, documents, metadata, where pre-filter, paging, tombstones,
persistence). CPU runs inject the TEST-ONLY oracle engine; the same assertions run on the GPU engine in
test_gpu_collection.py."""
import numpy as np
import pytest

from rag_dpo_amd import synth
from rag_dpo_amd.collection import Collection, DuplicateIDError, NotFoundError, PersistentClient
from rag_dpo_amd.where import evaluate, validate_where, Column, pack_bits

from oracle_engine import factory as oracle_factory

NAT = ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE"]


def fill(col, n=500, dim=64, seed=0):
    emb = synth.make_corpus(n, dim)
    ids = [f"chunk_{i}" for i in range(n)]
    docs = [f"doc {i}" for i in range(n)]
    metas = [{"document_path": f"p{i % 20}", "chunk_nature": NAT[i % 4], "chunk_index": i,
              "source": "ENTREPRISE" if i % 10 == 0 else "CNIL", "confidence": 0.5 + (i % 5) / 10,
              "is_priority": i % 3 == 0, **({"tag_rh": True} if i % 10 == 0 and i % 20 == 0 else {})}
             for i in range(n)]
    for a in range(0, n, 100):   # the reference's indexer adds in batches of 100
        col.add(ids=ids[a:a + 100], documents=docs[a:a + 100], embeddings=emb[a:a + 100].tolist(), metadatas=metas[a:a + 100])
    return emb, ids, docs, metas


def run_collection_contract(factory):
    col = Collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"}, engine_factory=factory)
    assert col.count() == 0
    r = col.query(query_embeddings=[[0.0] * 64], n_results=3)
    assert r["ids"] == [[]]
    emb, ids, docs, metas = fill(col)
    assert col.count() == 500

    # query as the reference's retriever calls it (retriever.py:215-220)
    q = synth.make_queries(1, 64, emb)[0].tolist()
    res = col.query(query_embeddings=[q], n_results=50, where=None, include=["documents", "metadatas", "distances"])
    assert set(res) >= {"ids", "documents", "metadatas", "distances"}
    assert len(res["ids"]) == 1 and len(res["ids"][0]) == 50
    d = res["distances"][0]
    assert all(isinstance(x, float) for x in d) and d == sorted(d) and 0 <= d[0] <= 2
    for cid, doc, meta in zip(res["ids"][0], res["documents"][0], res["metadatas"][0]):
        i = int(cid.split("_")[1])
        assert doc == docs[i] and meta == metas[i]
    # exactness against numpy on the normalised vectors
    en = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    qn = np.asarray(q) / np.linalg.norm(q)
    best = np.argsort(-(en @ qn), kind="stable")[:50]
    assert [f"chunk_{i}" for i in best[:10]] == res["ids"][0][:10]
    assert abs((1 - float(en[best[0]] @ qn)) - d[0]) < 1e-5

    # no include / no where (create_chromadb_index.py:405-408): default include has docs+metas+distances
    res2 = col.query(query_embeddings=[q], n_results=10)
    assert res2["ids"][0] == res["ids"][0][:10] and res2["documents"] is not None and res2["embeddings"] is None

    # where = pre-filter: still n_results hits, all passing (create_chromadb_index.py:435-439)
    res3 = col.query(query_embeddings=[q], n_results=3, where={"chunk_nature": "GUIDE"})
    assert len(res3["ids"][0]) == 3 and all(m["chunk_nature"] == "GUIDE" for m in res3["metadatas"][0])
    # the enterprise filter the pipeline builds (pipeline.py:35-71)
    w = {"$and": [{"chunk_nature": {"$in": ["GUIDE", "DOCTRINE"]}},
                  {"$or": [{"source": {"$ne": "ENTREPRISE"}}, {"tag_rh": True}]}]}
    res4 = col.query(query_embeddings=[q], n_results=400, where=w)
    exp = [i for i in range(500) if NAT[i % 4] in ("GUIDE", "DOCTRINE") and (i % 10 != 0 or i % 20 == 0)]
    assert sorted(res4["ids"][0]) == sorted(f"chunk_{i}" for i in exp)   # fewer than n_results rows pass
    assert res4["distances"][0] == sorted(res4["distances"][0])

    # batched multi-query == the same queries one at a time
    qs = synth.make_queries(4, 64, emb)
    many = col.query(query_embeddings=qs.tolist(), n_results=7)
    for b in range(4):
        one = col.query(query_embeddings=[qs[b].tolist()], n_results=7)
        assert one["ids"][0] == many["ids"][b] and one["distances"][0] == many["distances"][b]

    # get: paging as bm25_index.py:211-215 does
    page = col.get(limit=120, offset=100, include=["documents", "metadatas"])
    assert page["ids"] == ids[100:220] and page["documents"] == docs[100:220] and page["metadatas"][0] == metas[100]
    assert col.get(include=[])["ids"] == ids and col.get(include=[])["documents"] is None
    assert len(col.get(where={"source": "CNIL"}, limit=100000)["ids"]) == 450
    g = col.get(ids=["chunk_7", "nope", "chunk_3"], include=["embeddings"])
    assert g["ids"] == ["chunk_3", "chunk_7"] and np.allclose(g["embeddings"], en[[3, 7]], atol=1e-6)

    # update(ids=, metadatas=) merges (tag_all_chunks.py:215)
    col.update(ids=["chunk_1", "missing"], metadatas=[{"tags": "a,b"}, {"tags": "x"}])
    assert col.get(ids=["chunk_1"])["metadatas"][0] == {**metas[1], "tags": "a,b"}

    # add of an existing id is ignored; duplicates inside a call raise
    col.add(ids=["chunk_0", "new_1"], embeddings=emb[:2].tolist(), documents=["zzz", "n1"], metadatas=[{"a": 1}, {"a": 2}])
    assert col.count() == 501 and col.get(ids=["chunk_0"])["documents"] == ["doc 0"]
    with pytest.raises(DuplicateIDError):
        col.add(ids=["d", "d"], embeddings=emb[:2].tolist())
    with pytest.raises(ValueError):
        col.add(ids=["e1"], embeddings=[[1.0] * 32])            # wrong dimension
    with pytest.raises(ValueError):
        col.add(ids=["e2"], embeddings=[[float("nan")] * 64])
    with pytest.raises(ValueError):
        col.query(query_embeddings=[q], n_results=0)
    with pytest.raises(ValueError):
        col.query(query_embeddings=[q], n_results=3, where={"a": 1, "b": 2})
    assert col.count() == 501

    # delete(ids=) -> tombstones are never returned; heavy deletion compacts
    col.delete(ids=res["ids"][0][:5] + ["unknown"])
    assert col.count() == 496
    res5 = col.query(query_embeddings=[q], n_results=45)
    assert res5["ids"][0] == res["ids"][0][5:50]
    col.delete(ids=[f"chunk_{i}" for i in range(100, 400)])
    assert col.count() < 250
    res6 = col.query(query_embeddings=[q], n_results=500)
    alive = set(col.get(include=[])["ids"])
    assert set(res6["ids"][0]) == alive and len(res6["ids"][0]) == col.count()
    col.delete(where={"chunk_nature": "SANCTION"})
    assert all(m.get("chunk_nature") != "SANCTION" for m in col.get()["metadatas"])
    return col


def test_collection_contract_cpu():
    run_collection_contract(oracle_factory)


def test_where_grammar():
    n = 8
    cols = {"s": Column(n), "i": Column(n), "f": Column(n), "b": Column(n)}
    vals = [("a", 1, 1.0, True), ("b", 2, 2.5, False), ("a", 3, -1.0, True), (None, None, None, None),
            ("c", 1, 1.0, False), ("a", 2, 0.0, True), ("b", 0, 9.0, None), ("", -5, 1.0, False)]
    for r, (s, i, f, b) in enumerate(vals):
        cols["s"].set(r, s); cols["i"].set(r, i); cols["f"].set(r, f); cols["b"].set(r, b)
    ev = lambda w: np.flatnonzero(evaluate(w, cols, n)).tolist()
    assert ev({"s": "a"}) == [0, 2, 5]
    assert ev({"s": {"$eq": "a"}}) == [0, 2, 5]
    assert ev({"s": {"$ne": "a"}}) == [1, 3, 4, 6, 7]           # a missing key passes $ne
    assert ev({"s": {"$in": ["b", "c"]}}) == [1, 4, 6]
    assert ev({"s": {"$nin": ["b", "c"]}}) == [0, 2, 3, 5, 7]
    assert ev({"i": 1}) == [0, 4] and ev({"f": 1.0}) == [0, 4, 7]
    assert ev({"i": 1.0}) == []                                  # typed comparison: float never matches an int
    assert ev({"b": True}) == [0, 2, 5] and ev({"b": 1}) == []
    assert ev({"i": {"$gte": 2}}) == [1, 2, 5] and ev({"f": {"$lt": 1.0}}) == [2, 5]
    assert ev({"$or": [{"s": "c"}, {"$and": [{"i": {"$gt": 1}}, {"b": True}]}]}) == [2, 4, 5]
    assert ev({"nokey": "x"}) == [] and ev({"nokey": {"$ne": "x"}}) == list(range(n))
    assert evaluate(None, cols, n) is None and evaluate({}, cols, n) is None
    for bad in ({"a": 1, "b": 2}, {"$xor": [{"a": 1}, {"b": 2}]}, {"a": {"$in": []}}, {"$and": [{"a": 1}]},
                {"a": {"$gt": "x"}}, {"a": [1, 2]}, {"a": {"$in": [1, "x"]}}, "nope"):
        with pytest.raises(ValueError):
            validate_where(bad); evaluate(bad, cols, n)
    m = np.zeros(70, bool); m[[0, 31, 32, 69]] = True
    bits = pack_bits(m)
    assert bits.tolist() == [1 | (1 << 31), 1, 1 << 5]


def test_persistent_client_roundtrip(tmp_path):
    cl = PersistentClient(path=str(tmp_path / "db"), engine_factory=oracle_factory)
    with pytest.raises(NotFoundError):
        cl.get_collection("rag_dpo_chunks")
    col = cl.create_collection(name="rag_dpo_chunks", metadata={"description": "x", "hnsw:space": "cosine"})
    emb, ids, docs, metas = fill(col, n=300)
    col.delete(ids=ids[:10])
    with pytest.raises(ValueError):
        cl.create_collection(name="rag_dpo_chunks")
    assert cl.get_or_create_collection("rag_dpo_chunks") is col
    q = synth.make_queries(1, 64, emb).tolist()
    before = col.query(query_embeddings=q, n_results=20)
    cl.persist()
    cl2 = PersistentClient(path=str(tmp_path / "db"), engine_factory=oracle_factory)
    col2 = cl2.get_collection("rag_dpo_chunks")
    assert col2.count() == 290 and col2.metadata["hnsw:space"] == "cosine"
    after = col2.query(query_embeddings=q, n_results=20)
    assert after["ids"] == before["ids"] and after["documents"] == before["documents"]
    assert np.allclose(after["distances"], before["distances"], atol=1e-6)   # re-normalising stored unit rows
    cl2.delete_collection("rag_dpo_chunks")
    assert cl2.list_collections() == []
    assert PersistentClient(path=str(tmp_path / "db"), engine_factory=oracle_factory).list_collections() == []


def test_default_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rag_dpo_amd._lib import RdxUnavailable
    col = Collection("c")
    with pytest.raises(RdxUnavailable):
        col.add(ids=["a"], embeddings=[[1.0] * 64])


def test_import_collection_pages_through_a_chroma_shaped_source(tmp_path, factory=oracle_factory):
    from rag_dpo_amd.collection import PersistentClient, import_collection
    src = PersistentClient(path=None, engine_factory=factory).create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb, ids, docs, metas = fill(src, n=230, dim=32)
    src.delete(ids=ids[5:9])
    dst_client = PersistentClient(path=str(tmp_path / "db"), engine_factory=factory)
    dst = import_collection(src, dst_client, page=64)
    a, b = src.get(include=["documents", "metadatas"]), dst.get(include=["documents", "metadatas"])
    assert a["ids"] == b["ids"] and a["documents"] == b["documents"] and a["metadatas"] == b["metadatas"]
    q = emb[:3].tolist()
    assert src.query(query_embeddings=q, n_results=7)["ids"] == dst.query(query_embeddings=q, n_results=7)["ids"]
    again = PersistentClient(path=str(tmp_path / "db"), engine_factory=factory).get_collection("rag_dpo_chunks")
    assert again.get(include=[])["ids"] == a["ids"]
