#!/usr/bin/env python3
"""Captures tests/golden/indexer_golden.json by running the reference's OWN ChromaDBIndexer
(/root/reference/src/processing/create_chromadb_index.py, imported here, never copied) with `chromadb` bound to this
repo's drop-in (`rag_dpo_amd.collection.PersistentClient`, oracle engine: no GPU in the build container) and a
deterministic embedder in place of BGE-M3. `sentence_transformers` (absent here) is stubbed so the reference's
embedding_provider module imports; its model is never loaded. Runs only in the build container; the JSON is the fixture.

    python tests/golden/make_indexer_golden.py
"""
import functools
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import fixture_world as W  # noqa: E402
import indexer_world as IW  # noqa: E402
from oracle_engine import factory  # noqa: E402
from rag_dpo_amd import collection as rdx_collection  # noqa: E402

chroma = types.ModuleType("chromadb")
chroma.PersistentClient = functools.partial(rdx_collection.PersistentClient, engine_factory=factory)
chroma_cfg = types.ModuleType("chromadb.config")
chroma_cfg.Settings = lambda **kw: dict(kw)
chroma.config = chroma_cfg
sys.modules["chromadb"], sys.modules["chromadb.config"] = chroma, chroma_cfg
st = types.ModuleType("sentence_transformers")
st.SentenceTransformer = type("SentenceTransformer", (), {})
sys.modules["sentence_transformers"] = st
sys.path.insert(0, "/root/reference")
from src.processing.create_chromadb_index import ChromaDBIndexer  # noqa: E402  (the reference)


class Recorder:
    def __init__(self, col):
        self.col, self.calls = col, []

    def _rec(self, name, kw, res):
        kw = {k: v for k, v in kw.items() if k != "query_embeddings"}
        self.calls.append({"call": name, "kwargs": kw, "ids": res["ids"]})
        return res

    def query(self, **kw):
        return self._rec("query", kw, self.col.query(**kw))

    def get(self, **kw):
        return self._rec("get", kw, self.col.get(**kw))

    def __getattr__(self, name):
        return getattr(self.col, name)


def dump(col):
    g = col.get(include=["documents", "metadatas", "embeddings"])
    return {"ids": g["ids"], "documents": g["documents"], "metadatas": g["metadatas"],
            "embeddings_head": [[float(x) for x in e[:4]] for e in g["embeddings"]]}


def main():
    out = {"generator": "tests/golden/make_indexer_golden.py driving /root/reference/src/processing/create_chromadb_index.py"}
    with tempfile.TemporaryDirectory() as root:
        IW.write_project(root)
        ix = ChromaDBIndexer(project_root=root)
        ix.embedding_provider = IW.FlakyEmbedder()
        out["url_cache"] = dict(ix.url_cache)
        ix.init_chromadb(mode="reset")
        chunks = ix.load_chunks()
        out["n_loaded"] = len(chunks)
        ix.index_chunks(chunks, batch_size=IW.BATCH)
        out["stats_after_reset"] = dict(ix.stats)
        out["records_after_reset"] = dump(ix.collection)
        rec = Recorder(ix.collection)
        ix.collection = rec
        ix.verify_index()
        out["verify_calls"] = rec.calls
        out["helpers"] = [{"path": p, "source": ix._detect_source(p), "source_type": ix._detect_source_type(p),
                           "is_priority": ix._is_priority_source(p), "url": ix._get_url(p)} for p in IW.HELPER_PATHS]
        # second run over the same store in 'update' mode: the existing ids are read back
        ix2 = ChromaDBIndexer(project_root=root)
        ix2.embedding_provider = W.HashEmbedder()
        ix2.init_chromadb(mode="update")
        out["update_existing_ids"] = sorted(ix2.existing_ids)
        new = [c for c in IW.extra_chunks() if c.get("chunk_id") not in ix2.existing_ids]
        ix2.index_chunks(new, batch_size=IW.BATCH)
        out["stats_after_update"] = dict(ix2.stats)
        out["records_after_update"] = dump(ix2.collection)
    with open(os.path.join(HERE, "indexer_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print("wrote", out["n_loaded"], "chunks;", out["stats_after_reset"], out["stats_after_update"])


if __name__ == "__main__":
    main()
