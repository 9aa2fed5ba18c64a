"""Ingest side: this repo's ChromaDBIndexer must leave exactly the records the reference's own
indexer left when it was run against this repo's PersistentClient (tests/golden/make_indexer_golden.py captured them in
the build container; only the JSON travels). Engine = the TEST-ONLY oracle engine (CPU suite); the same flow on the HIP
engine is tests/test_gpu_collection.py."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

import fixture_world as W  # noqa: E402
import indexer_world as IW  # noqa: E402
from oracle_engine import factory  # noqa: E402
from rag_dpo_amd import indexer as ix_mod  # noqa: E402
from rag_dpo_amd.collection import PersistentClient  # noqa: E402

with open(os.path.join(HERE, "golden", "indexer_golden.json"), encoding="utf-8") as f:
    GOLD = json.load(f)


def dump(col):
    g = col.get(include=["documents", "metadatas", "embeddings"])
    return {"ids": g["ids"], "documents": g["documents"], "metadatas": g["metadatas"],
            "embeddings_head": [[float(x) for x in e[:4]] for e in g["embeddings"]]}


def same_records(got, want):
    assert got["ids"] == want["ids"]
    assert got["documents"] == want["documents"]
    assert got["metadatas"] == want["metadatas"]
    np.testing.assert_allclose(np.array(got["embeddings_head"]), np.array(want["embeddings_head"]), rtol=0, atol=1e-6)


def test_helpers_match_reference(tmp_path):
    _, manifest = IW.write_project(str(tmp_path))
    cache = ix_mod.load_url_cache(manifest)
    assert cache == GOLD["url_cache"]
    for h in GOLD["helpers"]:
        p = h["path"]
        assert ix_mod.detect_source(p) == h["source"], p
        assert ix_mod.detect_source_type(p) == h["source_type"], p
        assert ix_mod.is_priority_source(p) == h["is_priority"], p
        assert cache.get(p.replace("\\", "/"), p) == h["url"], p


def test_reset_then_update_leaves_the_reference_records(tmp_path):
    run_reset_then_update(tmp_path, factory)


def run_reset_then_update(tmp_path, factory):
    chunks_file, manifest = IW.write_project(str(tmp_path))
    store = str(tmp_path / "data" / "vectordb" / "chromadb")
    client = PersistentClient(path=store, engine_factory=factory)
    ix = ix_mod.ChromaDBIndexer(client, IW.FlakyEmbedder(), url_cache=ix_mod.load_url_cache(manifest))
    ix.init_chromadb("reset")
    chunks = ix.load_chunks(chunks_file)
    assert len(chunks) == GOLD["n_loaded"]
    ix.index_chunks(chunks, batch_size=IW.BATCH)
    assert ix.stats == GOLD["stats_after_reset"]
    same_records(dump(ix.collection), GOLD["records_after_reset"])

    # verify_index asks the collection the reference's questions and gets the reference's answers
    v = ix.verify_index()
    calls = GOLD["verify_calls"]
    assert v["count"] == len(GOLD["records_after_reset"]["ids"])
    assert v["top_ids"] == calls[0]["ids"][0] and calls[0]["kwargs"] == {"n_results": 10}
    assert v["guide_ids"] == calls[1]["ids"][0] and calls[1]["kwargs"] == {"n_results": 3, "where": {"chunk_nature": "GUIDE"}}
    assert set(v["guide_natures"]) <= {"GUIDE"}
    by = {json.dumps(c["kwargs"]["where"], sort_keys=True): len(c["ids"]) for c in calls[2:]}
    for s, n in v["by_source"].items():
        assert by[json.dumps({"source": s})] == n
    for s, n in v["by_nature"].items():
        assert by[json.dumps({"chunk_nature": s})] == n

    # a second process opens the same store (nobody called persist, as in the reference) in 'update' mode
    client2 = PersistentClient(path=store, engine_factory=factory)
    ix2 = ix_mod.ChromaDBIndexer(client2, W.HashEmbedder())
    ix2.init_chromadb("update")
    assert sorted(ix2.existing_ids) == GOLD["update_existing_ids"]
    new = [c for c in IW.extra_chunks() if c.get("chunk_id") not in ix2.existing_ids]
    ix2.index_chunks(new, batch_size=IW.BATCH)
    assert ix2.stats == GOLD["stats_after_update"]
    same_records(dump(ix2.collection), GOLD["records_after_update"])

    # snapshot + reopen: same records again
    client2.persist()
    assert not os.path.exists(os.path.join(store, ix_mod.COLLECTION_NAME, "journal.jsonl"))
    client3 = PersistentClient(path=store, engine_factory=factory)
    same_records(dump(client3.get_collection(ix_mod.COLLECTION_NAME)), GOLD["records_after_update"])


def test_journal_replays_updates_and_deletes(tmp_path):
    store = str(tmp_path / "db")
    c = PersistentClient(path=store, engine_factory=factory).create_collection("rag_dpo_chunks", metadata={"hnsw:space": "cosine"})
    emb = W.corpus()[:40]
    c.add(ids=[f"c{i}" for i in range(40)], embeddings=emb.tolist(), documents=[f"d{i}" for i in range(40)],
          metadatas=[{"chunk_nature": W.NAT[i % 4], "n": i} for i in range(40)])
    c.update(ids=["c3", "c5"], metadatas=[{"tag_rh": True}, {"n": 500}])
    c.update(ids=["c7"], embeddings=[emb[8].tolist()], documents=["d7 bis"])
    c.delete(ids=["c0", "c1"])
    c.delete(where={"chunk_nature": "SANCTION"})
    want = dump(c)
    with open(os.path.join(store, "rag_dpo_chunks", "journal.jsonl"), "a", encoding="utf-8") as f:
        f.write('{"op": "add", "ids": ["torn')   # a writer killed mid-line: that op never committed
    c2 = PersistentClient(path=store, engine_factory=factory).get_collection("rag_dpo_chunks")
    same_records(dump(c2), want)
    assert c2.count() == c.count() == 40 - 2 - 10   # c0, c1 and the ten SANCTION rows are gone
