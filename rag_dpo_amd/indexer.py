"""Corpus ingest — the write side of the hot path.

Mirror of the reference's `ChromaDBIndexer` (src/processing/create_chromadb_index.py): same collection name and
cosine space (:100-106), same document text `heading\\n\\ntext` (:325-330), same 18 scalar metadata fields and defaults
(:339-360), same batches of 100 chunks -> `embedding_provider.embed(documents)` -> `collection.add(...)` (:365-379), same
error accounting (a failing batch is counted and skipped, :367-370, :383-385), same `reset / append / update` modes
(:93-130) and the same verification queries (:389-486). What changes underneath: `embed` and `add` end in librdx
(K1 normalise on the device, rows resident in HBM) instead of sentence-transformers + chromadb.

Pinned by tests/golden/indexer_golden.json: the reference's own indexer, imported and run against this repo's
`PersistentClient`, must leave exactly the records this class leaves (tests/golden/make_indexer_golden.py).
"""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Dict, List, Optional

logger = logging.getLogger(__name__)

COLLECTION_NAME = "rag_dpo_chunks"                      # reference src/utils/paths.py:52
COLLECTION_METADATA = {"description": "Chunks RGPD/CNIL avec classification hybride", "hnsw:space": "cosine"}


def detect_source(doc_path: str) -> str:
    """'ENTREPRISE' for company documents, else 'CNIL' (reference :226-241)"""
    p = doc_path.lower()
    return "ENTREPRISE" if ("entreprise" in p or "custom" in p or "internal" in p) else "CNIL"


def detect_source_type(doc_path: str) -> str:
    """reference :243-264"""
    ext = Path(doc_path).suffix.lower()
    return {".html": "html", ".htm": "html", ".pdf": "pdf", ".docx": "docx", ".doc": "docx", ".xlsx": "xlsx", ".xls": "xlsx",
            ".odt": "odt", ".ods": "odt"}.get(ext, "unknown")


def is_priority_source(doc_path: str) -> bool:
    """reference :266-288"""
    p = doc_path.lower()
    return any(w in p for w in ("entreprise", "custom", "internal", "template", "modele", "modèle", "politique", "policy"))


def load_url_cache(keep_manifest_path) -> Dict[str, str]:
    """{normalised file_path: url} from keep_manifest.json; a document without its own url falls back to the page that
    links to it (reference :155-206). A missing or unreadable manifest yields an empty cache."""
    cache: Dict[str, str] = {}
    try:
        with open(keep_manifest_path, "r", encoding="utf-8") as f:
            keep = json.load(f)
        for key in ("html", "pdfs", "docs"):
            for item in keep.get(key, []):
                meta = item.get("metadata", {})
                file_path = meta.get("file_path", "")
                if not file_path:
                    continue
                url = meta.get("url", "") or item.get("url", "")
                parent = item.get("parent_url", "") or meta.get("source_url", "")
                if url or parent:
                    cache[file_path.replace("\\", "/")] = url or parent
    except Exception as e:   # noqa: BLE001
        logger.warning(f"keep_manifest not loaded: {e}")
    return cache


def chunk_document(chunk: Dict) -> str:
    """text that is embedded and stored (reference :321-330)"""
    text, heading = chunk.get("text", ""), chunk.get("heading", "")
    return f"{heading}\n\n{text}" if heading else text


def chunk_metadata(chunk: Dict, url_cache: Optional[Dict[str, str]] = None) -> Dict:
    """the 18 scalar fields the `where` filters run on (reference :332-360)"""
    text, heading = chunk.get("text", ""), chunk.get("heading", "")
    doc_path = chunk.get("document_path", "")
    source_url = chunk.get("source_url", "") or (url_cache or {}).get(doc_path.replace("\\", "/"), doc_path)
    return {
        "document_id": chunk.get("document_id", ""),
        "document_path": doc_path,
        "document_nature": chunk.get("document_nature", "GUIDE"),
        "chunk_nature": chunk.get("chunk_nature", "GUIDE"),
        "chunk_index": chunk.get("chunk_index", "OPERATIONNEL"),
        "heading": heading[:200] if heading else "",
        "page_info": chunk.get("page_info", ""),
        "confidence": chunk.get("confidence", 0.5),
        "method": chunk.get("method", "unknown"),
        "word_count": len(text.split()),
        "sectors": ",".join(chunk.get("sectors", [])),
        "file_type": chunk.get("file_type", detect_source_type(doc_path)),
        "title": chunk.get("title", "")[:300],
        "source": detect_source(doc_path),
        "source_type": detect_source_type(doc_path),
        "is_priority": is_priority_source(doc_path),
        "source_url": source_url,
        "parent_url": chunk.get("parent_url", ""),
    }


class ChromaDBIndexer:
    def __init__(self, client, embedding_provider, url_cache: Optional[Dict[str, str]] = None, device_embeddings: bool = False):
        """device_embeddings=True: batches go `embedding_provider.embed_device(texts)` -> `collection.add(embeddings=<CUDA
        tensor>)`, i.e. encoder output -> K1 -> HBM without ever becoming Python floats. The default keeps
        the reference's list-of-lists hand-over."""
        self.chroma_client = client
        self.embedding_provider = embedding_provider
        self.device_embeddings = device_embeddings and hasattr(embedding_provider, "embed_device")
        self.url_cache = dict(url_cache or {})
        self.collection = None
        self.existing_ids = set()
        self.stats = {"chunks_loaded": 0, "chunks_indexed": 0, "errors": 0}

    def init_chromadb(self, mode: str = "reset"):
        """reference :70-130"""
        if mode == "reset":
            try:
                self.chroma_client.delete_collection(name=COLLECTION_NAME)
            except Exception:   # noqa: BLE001  (reference: bare except, a missing collection is fine)
                pass
            self.collection = self.chroma_client.create_collection(name=COLLECTION_NAME, metadata=dict(COLLECTION_METADATA))
        elif mode in ("append", "update"):
            try:
                self.collection = self.chroma_client.get_collection(name=COLLECTION_NAME)
                if mode == "update":
                    self.existing_ids = set(self.collection.get(include=[])["ids"])
            except Exception:   # noqa: BLE001
                self.collection = self.chroma_client.create_collection(name=COLLECTION_NAME, metadata=dict(COLLECTION_METADATA))
                self.existing_ids = set()
        else:
            raise ValueError(f"unknown mode {mode!r}")
        return self.collection

    def load_chunks(self, chunks_file) -> List[Dict]:
        """JSONL -> list of chunk dicts; unparsable lines are counted, not fatal (reference :132-153)"""
        path = Path(chunks_file)
        if not path.exists():
            logger.error(f"{path} not found")
            return []
        chunks = []
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                try:
                    chunks.append(json.loads(line))
                except Exception as e:   # noqa: BLE001
                    logger.warning(f"bad chunk line: {e}")
                    self.stats["errors"] += 1
        self.stats["chunks_loaded"] = len(chunks)
        return chunks

    def generate_embeddings(self, texts: List[str]) -> List[List[float]]:
        """reference :290-298: an embedder failure yields [] and the batch is skipped"""
        try:
            if self.device_embeddings:
                return self.embedding_provider.embed_device(texts)
            return self.embedding_provider.embed(texts)
        except Exception as e:   # noqa: BLE001
            logger.error(f"embedding error: {e}")
            return []

    def _prepare_batch(self, chunks: List[Dict], i: int, batch_size: int):
        batch = chunks[i:i + batch_size]
        ids = [c.get("chunk_id", f"chunk_{i}") for c in batch]
        documents = [chunk_document(c) for c in batch]
        metadatas = [chunk_metadata(c, self.url_cache) for c in batch]
        return batch, ids, documents, metadatas, self.generate_embeddings(documents)

    def index_chunks(self, chunks: List[Dict], batch_size: int = 100):
        """reference :300-387. Same batches, same order of `collection.add` calls, same error accounting. With device embeddings the
        NEXT batch is prepared — documents, metadata, tokenising, the forward's launches — on a worker thread while this batch's
        `collection.add` waits for the GPU (it synchronises the device before K1 reads the embeddings): the host's ~10 ms per batch
        of 100 chunks disappear behind the GPU's ~20 (profiles/r04/ingest_host_profile.txt)."""
        starts = list(range(0, len(chunks), batch_size))
        pool = None
        if self.device_embeddings and len(starts) > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)
        try:
            nxt = pool.submit(self._prepare_batch, chunks, starts[0], batch_size) if pool else None
            for j, i in enumerate(starts):
                if pool:
                    batch, ids, documents, metadatas, embeddings = nxt.result()
                    nxt = pool.submit(self._prepare_batch, chunks, starts[j + 1], batch_size) if j + 1 < len(starts) else None
                else:
                    batch, ids, documents, metadatas, embeddings = self._prepare_batch(chunks, i, batch_size)
                if embeddings is None or len(embeddings) != len(documents):
                    self.stats["errors"] += len(batch)
                    continue
                try:
                    self.collection.add(ids=ids, documents=documents, embeddings=embeddings, metadatas=metadatas)
                    self.stats["chunks_indexed"] += len(batch)
                except Exception as e:   # noqa: BLE001
                    logger.error(f"indexing error in batch {i // batch_size}: {e}")
                    self.stats["errors"] += len(batch)
        finally:
            if pool:
                pool.shutdown(wait=True)

    def verify_index(self) -> Dict:
        """the reference's three checks (:389-486), returned instead of only logged"""
        count = self.collection.count()
        q = self.embedding_provider.embed(["Comment faire une AIPD ?"])[0]
        r1 = self.collection.query(query_embeddings=[q], n_results=10)
        seen, top_docs = set(), []
        for meta in r1["metadatas"][0]:
            p = meta.get("document_path", "")
            if p not in seen:
                seen.add(p)
                top_docs.append(p)
                if len(top_docs) >= 3:
                    break
        r2 = self.collection.query(query_embeddings=[q], n_results=3, where={"chunk_nature": "GUIDE"})
        by = lambda w: len(self.collection.get(where=w, limit=100000)["ids"])   # noqa: E731
        return {
            "count": count, "top_ids": r1["ids"][0], "top_documents": top_docs, "guide_ids": r2["ids"][0],
            "guide_natures": [m.get("chunk_nature") for m in r2["metadatas"][0]],
            "by_source": {s: by({"source": s}) for s in ("CNIL", "ENTREPRISE")},
            "by_nature": {n: by({"chunk_nature": n}) for n in ("DOCTRINE", "GUIDE", "SANCTION", "TECHNIQUE")},
        }

    def run(self, chunks_file, mode: str = "reset", batch_size: int = 100) -> Dict:
        self.init_chromadb(mode)
        chunks = self.load_chunks(chunks_file)
        if mode == "update":
            chunks = [c for c in chunks if c.get("chunk_id") not in self.existing_ids]
        self.index_chunks(chunks, batch_size=batch_size)
        if hasattr(self.chroma_client, "persist"):
            self.chroma_client.persist()
        return dict(self.stats)
