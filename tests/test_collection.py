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


# ---- crash consistency of the persisted store (journal + snapshot generations) ---------------------------------

def _open(tmp_path):
    return PersistentClient(path=str(tmp_path / "db"), engine_factory=oracle_factory)


def _state(col):
    g = col.get(include=["documents", "metadatas", "embeddings"])
    return g["ids"], g["documents"], g["metadatas"], np.asarray(g["embeddings"])


def test_journal_survives_a_kill_between_its_two_writes(tmp_path):
    """a writer killed after the vectors of an op reached journal.f32 but before its jsonl line: the orphan floats must
    not shift the vectors of later ops (each record names its own byte range) and vanish at the next open"""
    import os
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = synth.make_corpus(40, 64)
    col.add(ids=[f"a{i}" for i in range(10)], embeddings=emb[:10].tolist(), documents=[f"d{i}" for i in range(10)])
    d = col._dir
    jf = os.path.join(d, col._cur_names()["jf"])
    committed = os.path.getsize(jf)
    with open(jf, "ab") as f:                       # the killed writer's vectors: 7 rows that never got their line
        f.write(np.full((7, 64), 123.0, dtype=np.float32).tobytes())
    cl2 = _open(tmp_path)                           # next process: opens, appends its own op
    col2 = cl2.get_collection("rag_dpo_chunks")
    assert col2.count() == 10 and os.path.getsize(jf) == committed        # orphan floats truncated away
    col2.add(ids=[f"b{i}" for i in range(5)], embeddings=emb[10:15].tolist())
    col3 = _open(tmp_path).get_collection("rag_dpo_chunks")
    ids, _, _, e = _state(col3)
    assert ids == [f"a{i}" for i in range(10)] + [f"b{i}" for i in range(5)]
    ref = emb[:15] / np.linalg.norm(emb[:15], axis=1, keepdims=True)
    assert np.allclose(e, ref, atol=1e-6)           # b0..b4 carry THEIR vectors, not the orphan's
    import json
    recs = [json.loads(l) for l in open(os.path.join(d, col._cur_names()["jl"]))]
    assert all("f32_off" in r and r["f32_len"] == r["n_emb"] * r["dim"] * 4 for r in recs if r["n_emb"])


def test_journal_torn_last_line_is_dropped_and_the_next_record_starts_clean(tmp_path):
    import os
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = synth.make_corpus(30, 64)
    col.add(ids=["x0", "x1"], embeddings=emb[:2].tolist())
    col.update(ids=["x0"], metadatas=[{"tags": "t"}])
    jl = os.path.join(col._dir, col._cur_names()["jl"])
    with open(jl, "ab") as f:
        f.write(b'{"op": "add", "ids": ["torn0", "to')        # killed mid-line: no newline, invalid JSON
    col2 = _open(tmp_path).get_collection("rag_dpo_chunks")
    assert col2.get(include=[])["ids"] == ["x0", "x1"]
    col2.add(ids=["y0"], embeddings=emb[2:3].tolist(), metadatas=[{"k": 1}])
    col2.delete(ids=["x1"])
    col3 = _open(tmp_path).get_collection("rag_dpo_chunks")      # the new records did not fuse with the torn one
    assert col3.get(include=[])["ids"] == ["x0", "y0"]
    assert col3.get(ids=["x0"])["metadatas"][0] == {"tags": "t"} and col3.get(ids=["y0"])["metadatas"][0] == {"k": 1}


def test_persist_killed_before_or_after_its_commit_point(tmp_path):
    """persist() writes generation g+1 beside g and commits by replacing collection.json: a kill before the commit leaves g
    (+ its journal) in force, a kill right after leaves g+1 in force; the other generation's files go at the next open"""
    import os
    import shutil
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = synth.make_corpus(60, 64)
    col.add(ids=[f"a{i}" for i in range(20)], embeddings=emb[:20].tolist(), documents=[f"d{i}" for i in range(20)])
    cl.persist()                                                  # generation 1
    col.add(ids=[f"b{i}" for i in range(5)], embeddings=emb[20:25].tolist())
    col.delete(ids=["a3"])
    want = _state(col)
    d = col._dir
    before = str(tmp_path / "before")
    shutil.copytree(d, before)                                    # generation 1 + journal1
    cl.persist()                                                  # generation 2
    after_files = {fn: open(os.path.join(d, fn), "rb").read() for fn in os.listdir(d)}
    # (a) killed BEFORE the commit: gen-2 snapshot files exist, the header still names gen 1
    shutil.rmtree(d)
    shutil.copytree(before, d)
    for fn, blob in after_files.items():
        if fn.startswith("snap2."):
            open(os.path.join(d, fn), "wb").write(blob[: len(blob) // 2])    # half-written at that
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and (got[3] == want[3]).all()
    assert not [fn for fn in os.listdir(d) if fn.startswith("snap2.")]
    # (b) killed right AFTER the commit: header names gen 2, gen-1 files and journal1 are still lying around
    shutil.rmtree(d)
    shutil.copytree(before, d)
    for fn, blob in after_files.items():
        open(os.path.join(d, fn), "wb").write(blob)
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and (got[3] == want[3]).all()
    assert sorted(os.listdir(d)) == sorted(after_files)           # journal1 was NOT replayed on top of snapshot 2, and is gone


def test_failed_snapshot_leaves_the_store_on_its_old_generation(tmp_path, monkeypatch):
    """persist() that fails while writing generation g+1 (disk full in np.save): the object must stay on generation g, so that
    the writes acknowledged afterwards go to the journal the header names and are there at the next open"""
    import os
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = synth.make_corpus(40, 64)
    col.add(ids=[f"a{i}" for i in range(10)], embeddings=emb[:10].tolist(), documents=[f"d{i}" for i in range(10)])
    cl.persist()                                                  # generation 1
    col.add(ids=["b0"], embeddings=emb[10:11].tolist())
    gen = col._gen

    def full(*a, **kw):
        raise OSError(28, "No space left on device")
    monkeypatch.setattr(np, "save", full)
    with pytest.raises(OSError):
        cl.persist()
    monkeypatch.undo()
    assert col._gen == gen
    assert not [fn for fn in os.listdir(col._dir) if fn.startswith(f"snap{gen + 1}.") or fn.endswith(".tmp")]
    col.add(ids=["c0", "c1"], embeddings=emb[11:13].tolist(), documents=["x", "y"])   # acknowledged after the failure
    col.delete(ids=["a4"])
    want = _state(col)
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and (got[3] == want[3]).all()
    assert "c1" in got[0] and "a4" not in got[0]
    cl.persist()                                                  # and a later snapshot succeeds
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and (got[3] == want[3]).all()


def test_snapshot_failing_after_its_commit_point_keeps_the_new_generation(tmp_path, monkeypatch):
    """persist() whose header replace succeeded but whose directory fsync raised: collection.json on disk names
    generation g+1, so its files must NOT be removed, the object must follow the header, and writes acknowledged afterwards must
    be found again at the next open — a handler that deletes snap<g+1> there loses every row"""
    import os
    from rag_dpo_amd import collection as C
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = synth.make_corpus(40, 64)
    col.add(ids=[f"a{i}" for i in range(10)], embeddings=emb[:10].tolist(), documents=[f"d{i}" for i in range(10)])
    cl.persist()                                                  # generation 1
    col.add(ids=["b0"], embeddings=emb[10:11].tolist())
    gen = col._gen
    real, calls = C._fsync_dir, []

    def flaky(path):
        calls.append(path)
        if len(calls) == 2:                                       # 1st: before the header; 2nd: right after os.replace(collection.json)
            raise OSError(5, "Input/output error")
        return real(path)
    monkeypatch.setattr(C, "_fsync_dir", flaky)
    with pytest.raises(OSError):
        cl.persist()
    monkeypatch.undo()
    assert len(calls) == 2
    assert col._gen == gen + 1                                    # the object follows the header that is on disk
    have = set(os.listdir(col._dir))
    assert {f"snap{gen + 1}.embeddings.f32.npy", f"snap{gen + 1}.records.jsonl"} <= have
    col.add(ids=["c0", "c1"], embeddings=emb[11:13].tolist(), documents=["x", "y"])   # acknowledged after the failure
    col.delete(ids=["a4"])
    want = _state(col)
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and (got[3] == want[3]).all()
    assert "b0" in got[0] and "c1" in got[0] and "a4" not in got[0] and len(got[0]) == 12
    cl.persist()
    got = _state(_open(tmp_path).get_collection("rag_dpo_chunks"))
    assert got[0] == want[0] and (got[3] == want[3]).all()


def test_reload_keeps_stored_vectors_bit_for_bit(tmp_path):
    """snapshot rows are reloaded verbatim (engine.add_stored), not normalised a second time: distances before persist
    and after reopen are the same floats (the reference indexes in one process and serves from another)"""
    cl = _open(tmp_path)
    col = cl.create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb, ids, docs, metas = fill(col, n=300)
    q = synth.make_queries(3, 64, emb).tolist()
    before = col.query(query_embeddings=q, n_results=40)
    e0 = np.asarray(col.get(include=["embeddings"])["embeddings"])
    for _ in range(3):                                             # three persist/open cycles
        cl.persist()
        cl = _open(tmp_path)
        col = cl.get_collection("rag_dpo_chunks")
    after = col.query(query_embeddings=q, n_results=40)
    assert after["ids"] == before["ids"] and after["distances"] == before["distances"]
    assert (np.asarray(col.get(include=["embeddings"])["embeddings"]) == e0).all()


def test_bad_metadata_in_the_middle_of_a_batch_stores_nothing(tmp_path):
    """every metadata value is validated before the engine sees the batch: host rows and device rows cannot diverge"""
    col = Collection("c", engine_factory=oracle_factory)
    emb = synth.make_corpus(12, 64)
    col.add(ids=["a", "b"], embeddings=emb[:2].tolist(), metadatas=[{"n": 1}, {"n": 2}])
    for bad in ({"n": 2 ** 60}, {5: "x"}, {"n": [1, 2]}, {"n": 10 ** 400}):
        with pytest.raises((ValueError, OverflowError)):
            col.add(ids=["c", "d", "e"], embeddings=emb[2:5].tolist(), metadatas=[{"n": 3}, bad, {"n": 5}])
        assert col.count() == 2 and len(col._engine) == 2
    with pytest.raises(ValueError):
        col.update(ids=["a"], metadatas=[{"n": 2 ** 60}])
    col.add(ids=["c", "d"], embeddings=emb[2:4].tolist(), metadatas=[{"n": 3}, {"n": 4}])
    r = col.query(query_embeddings=emb[3:4].tolist(), n_results=1, where={"n": {"$gte": 3}})
    assert r["ids"] == [["d"]] and r["metadatas"][0][0] == {"n": 4}


def run_mask_cache(factory):
    """a filter seen before is not evaluated or uploaded again; any write drops the cached bitmaps; results never change"""
    col = Collection("c", engine_factory=factory)
    emb, ids, docs, metas = fill(col, n=400)
    q = synth.make_queries(2, 64, emb).tolist()
    w1 = {"$and": [{"chunk_nature": {"$in": ["GUIDE", "DOCTRINE"]}}, {"$or": [{"source": {"$ne": "ENTREPRISE"}}, {"tag_rh": True}]}]}
    w1_reordered = {"$and": [{"chunk_nature": {"$in": ["GUIDE", "DOCTRINE"]}}, {"$or": [{"source": {"$ne": "ENTREPRISE"}}, {"tag_rh": True}]}]}
    a = col.query(query_embeddings=q, n_results=30, where=w1)
    assert col.mask_cache_hits == 0 and len(col._mask_cache) == 1
    b = col.query(query_embeddings=q, n_results=30, where=w1_reordered)
    assert col.mask_cache_hits == 1 and a == b
    col.query(query_embeddings=q, n_results=30, where={"source": "CNIL"})
    assert len(col._mask_cache) == 2
    assert col.query(query_embeddings=q, n_results=30) == col.query(query_embeddings=q, n_results=30, where={})   # no filter: no mask
    assert len(col._mask_cache) == 2
    # writes invalidate: metadata update changes who passes
    first = a["ids"][0][0]
    col.update(ids=[first], metadatas=[{"chunk_nature": "SANCTION"}])
    assert not col._mask_cache
    c = col.query(query_embeddings=q, n_results=30, where=w1)
    assert first not in c["ids"][0] and c["ids"][0][:5] == a["ids"][0][1:6]
    # add: the new row passes the cached filter's successor, never a stale bitmap of the old row count
    col.add(ids=["new"], embeddings=[q[0]], metadatas=[{"chunk_nature": "GUIDE", "source": "CNIL"}])
    assert not col._mask_cache
    d = col.query(query_embeddings=q, n_results=5, where=w1)
    assert d["ids"][0][0] == "new"
    # delete: tombstones are part of every bitmap, also of the "no filter" one
    col.delete(ids=["new"])
    e = col.query(query_embeddings=q, n_results=5, where=w1)
    assert e["ids"][0] == c["ids"][0][:5]
    f1 = col.query(query_embeddings=q, n_results=5)
    f2 = col.query(query_embeddings=q, n_results=5)
    assert "new" not in f1["ids"][0] and f1 == f2 and col.mask_cache_hits >= 2
    # bounded
    for i in range(Collection._MASK_CACHE_MAX + 5):
        col.query(query_embeddings=q, n_results=1, where={"chunk_index": {"$gte": i}})
    assert len(col._mask_cache) <= Collection._MASK_CACHE_MAX
    return col


def test_mask_cache_cpu():
    run_mask_cache(oracle_factory)


def test_mask_cache_uses_resident_masks_when_the_engine_has_them():
    from oracle_engine import OracleEngine

    class ResidentEngine(OracleEngine):
        made, closed, used = 0, 0, 0

        def make_mask(self, bits):
            ResidentEngine.made += 1
            outer = self

            class M:
                def __init__(s):
                    s.bits, s.rows = np.array(bits), len(outer)

                def close(s):
                    ResidentEngine.closed += 1
            return M()

        def search(self, q, k, allow_bits=None, mask=None):
            if mask is not None:
                assert mask.rows == len(self)        # a mask never outlives a write
                ResidentEngine.used += 1
                allow_bits = mask.bits
            return super().search(q, k, allow_bits)

    run_mask_cache(lambda dim, device=0: ResidentEngine(dim, device))
    assert ResidentEngine.made > 5 and ResidentEngine.used > ResidentEngine.made
    assert ResidentEngine.closed >= ResidentEngine.made - Collection._MASK_CACHE_MAX
