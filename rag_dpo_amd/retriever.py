"""Host-side fusion either side of the dense hot path: the dense part of the reference's
`RAGRetriever.retrieve_candidates` / `.retrieve` (src/rag/retriever.py:156-470) restated so that the ≤ 4 queries
of one question are embedded in ONE batch and searched in ONE `collection.query` call (B = 4 instead of four
B = 1 round trips) while every per-query result, the rank fusion, the score merge and the
document de-duplication stay identical to the reference's.

Pinned by tests/golden/retriever_golden.json, captured by driving the IMPORTED reference retriever
(tests/golden/make_retriever_golden.py) with the same collection and embedder.
BM25 / summary pre-filter / LLM query expansion are out of scope: the expansion is an injected
callable, sparse rankings can be passed in as extra rankings.
"""
from __future__ import annotations

import logging
from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence


logger = logging.getLogger(__name__)


@dataclass
class RetrievedChunk:
    """same fields as reference src/rag/retriever.py:22-42"""
    chunk_id: str
    text: str
    document_path: str
    chunk_nature: str
    chunk_index: int
    confidence: str
    distance: float
    metadata: Dict[str, Any]
    bm25_score: float = 0.0
    semantic_score: float = 0.0
    hybrid_score: float = 0.0

    @property
    def similarity_score(self) -> float:
        return 1.0 / (1.0 + self.distance)          # reference retriever.py:39-42


@dataclass
class RetrievedDocument:
    """reference src/rag/retriever.py:45-63"""
    document_path: str
    chunks: List[RetrievedChunk]
    avg_similarity: float = 0.0
    primary_nature: str = ""

    def __post_init__(self):
        if self.chunks:
            self.avg_similarity = sum(c.similarity_score for c in self.chunks) / len(self.chunks)
            natures = [c.chunk_nature for c in self.chunks]
            self.primary_nature = max(set(natures), key=natures.count)
        else:
            self.avg_similarity = 0.0
            self.primary_nature = "UNKNOWN"


def reciprocal_rank_fusion(rankings: Sequence[Sequence[str]], k: int = 60,
                           weights: Optional[Sequence[float]] = None) -> Dict[str, float]:
    """score(id) = sum_i w_i / (k + rank_i + 1)   — reference src/rag/retriever.py:66-90"""
    if weights is None:
        weights = [1.0] * len(rankings)
    scores: Dict[str, float] = defaultdict(float)
    for ranking, weight in zip(rankings, weights):
        for rank, doc_id in enumerate(ranking):
            scores[doc_id] += weight / (k + rank + 1)
    return dict(scores)


def parse_query_results(results: Dict, b: int = 0) -> List[RetrievedChunk]:
    """reference src/rag/retriever.py:472-494 for the b-th query of a batched result"""
    chunks = []
    for chunk_id, text, metadata, distance in zip(results["ids"][b], results["documents"][b],
                                                  results["metadatas"][b], results["distances"][b]):
        metadata = metadata or {}
        chunks.append(RetrievedChunk(
            chunk_id=chunk_id, text=text, document_path=metadata.get("document_path", ""),
            chunk_nature=metadata.get("chunk_nature", "UNKNOWN"), chunk_index=metadata.get("chunk_index", 0),
            confidence=metadata.get("confidence", "unknown"), distance=distance, metadata=metadata))
    return chunks


class DenseRetriever:
    def __init__(self, collection, embedding_provider, query_expander: Optional[Callable[[str], List[str]]] = None,
                 query_preprocessor: Optional[Callable[[str], str]] = None, n_documents: int = 5,
                 n_chunks_per_doc: int = 3, fetch_multiplier: int = 10):
        self.collection = collection
        self.embedding_provider = embedding_provider
        self.query_expander = query_expander            # reference: QueryExpander.expand (LLM, out of scope)
        self.query_preprocessor = query_preprocessor    # reference: expand_query_with_acronyms (string op, out of scope)
        self.n_documents = n_documents
        self.n_chunks_per_doc = n_chunks_per_doc
        self.fetch_multiplier = fetch_multiplier

    # one batched device round trip for all the queries of a question
    def _dense(self, all_queries: List[str], n_fetch: int, where_filter):
        """-> per sub-query its chunks, or None where `collection.query` raised for that sub-query: the reference calls
        query once per sub-query inside a try and skips a failing one (src/rag/retriever.py:215-223, 380-388; an embed
        failure is NOT caught there, :212, :377). Here the sub-queries go down as ONE batch; only when that batched call
        raises (e.g. one NaN embedding rejects the batch) they are re-issued one at a time, so one bad sub-query costs
        exactly that sub-query, as in the reference."""
        vectors = self.embedding_provider.embed(all_queries)       # reference: one embed([q]) per query (:212, :377)
        include = ["documents", "metadatas", "distances"]
        try:
            results = self.collection.query(query_embeddings=vectors, n_results=n_fetch, where=where_filter, include=include)
            return [parse_query_results(results, b) for b in range(len(all_queries))]
        except Exception as e:
            logger.error("collection.query failed for the batch of %d sub-queries (%s): retrying one by one", len(all_queries), e)
        out: List[Optional[List[RetrievedChunk]]] = []
        for q_idx, v in enumerate(vectors):
            try:
                res = self.collection.query(query_embeddings=[v], n_results=n_fetch, where=where_filter, include=include)
                out.append(parse_query_results(res, 0))
            except Exception as e:
                logger.error("collection.query failed (%s): %s", "principale" if q_idx == 0 else f"expansion #{q_idx}", e)
                out.append(None)                                   # reference: `continue` (:221-223, :386-388)
        return out

    def _queries(self, query: str) -> List[str]:
        expanded = self.query_preprocessor(query) if self.query_preprocessor else query
        return list(self.query_expander(expanded)) if self.query_expander is not None else [expanded]

    def _fuse(self, per_query: List[List[RetrievedChunk]], extra_rankings=(), extra_weights=()):
        all_rankings, weights = [], []
        chunk_map: Dict[str, RetrievedChunk] = {}
        for q_idx, chunks in enumerate(per_query):
            if chunks is None:                                     # skipped sub-query: no ranking, no weight
                continue
            for c in chunks:
                c.semantic_score = c.similarity_score
            all_rankings.append([c.chunk_id for c in chunks])
            weights.append(2.0 if q_idx == 0 else 1.0)             # reference :209, :374
            for c in chunks:                                       # reference :246-256, :407-415
                old = chunk_map.get(c.chunk_id)
                if old is None:
                    chunk_map[c.chunk_id] = c
                else:
                    if c.distance < old.distance:
                        old.distance = c.distance
                    if c.semantic_score > old.semantic_score:
                        old.semantic_score = c.semantic_score
        all_rankings += [list(r) for r in extra_rankings]
        weights += list(extra_weights)
        if len(all_rankings) > 1:                                  # reference :293-300, :455-462
            rrf = reciprocal_rank_fusion(all_rankings, weights=weights)
            for cid, c in chunk_map.items():
                c.hybrid_score = rrf.get(cid, 0.0)
        else:
            for c in chunk_map.values():
                c.hybrid_score = c.semantic_score
        out = list(chunk_map.values())
        out.sort(key=lambda c: c.hybrid_score, reverse=True)
        return out

    def retrieve_candidates(self, query: str, n_candidates: int = 100, where_filter=None) -> List[RetrievedChunk]:
        """reference src/rag/retriever.py:312-470 without BM25 / summary pre-filter"""
        n_fetch = max(n_candidates, 50)
        per_query = self._dense(self._queries(query), n_fetch, where_filter)
        return self._fuse(per_query)[:n_candidates]

    def retrieve(self, query: str, where_filter=None, n_documents: Optional[int] = None,
                 n_chunks_per_doc: Optional[int] = None) -> List[RetrievedDocument]:
        """reference src/rag/retriever.py:156-310 without BM25 / summary pre-filter"""
        n_docs = n_documents or self.n_documents
        n_chunks = n_chunks_per_doc or self.n_chunks_per_doc
        per_query = self._dense(self._queries(query), n_docs * self.fetch_multiplier, where_filter)
        return deduplicate_by_document(self._fuse(per_query), n_docs, n_chunks)


def deduplicate_by_document(chunks: List[RetrievedChunk], n_documents: int, n_chunks_per_doc: int) -> List[RetrievedDocument]:
    """reference src/rag/retriever.py:539-578"""
    doc_chunks: Dict[str, List[RetrievedChunk]] = defaultdict(list)
    for c in chunks:
        doc_chunks[c.document_path].append(c)
    documents, seen_urls = [], set()
    for doc_path, lst in doc_chunks.items():
        ordered = sorted(lst, key=lambda c: c.hybrid_score if c.hybrid_score > 0 else c.similarity_score, reverse=True)
        selected = ordered[:n_chunks_per_doc]
        url = selected[0].metadata.get("source_url", "") if selected else ""
        if url:
            norm = url.lower().replace("https://", "").replace("http://", "").replace("www.", "")
            if norm in seen_urls:
                continue
            seen_urls.add(norm)
        documents.append(RetrievedDocument(document_path=doc_path, chunks=selected))
    documents.sort(key=lambda d: d.avg_similarity, reverse=True)
    return documents[:n_documents]


def build_enterprise_where_filter(base_filter: Optional[Dict] = None, enterprise_tags: Optional[List[str]] = None) -> Optional[Dict]:
    """the `where` the pipeline sends down to collection.query — reference src/rag/pipeline.py:35-71:
    no tags -> base filter unchanged; tags -> $or[source != ENTREPRISE, tag_X = True, ...] AND-ed with the base."""
    if not enterprise_tags:
        return base_filter
    source_filter = {"$or": [{"source": {"$ne": "ENTREPRISE"}}] + [{f"tag_{t}": True} for t in enterprise_tags]}
    if base_filter:
        return {"$and": [base_filter, source_filter]}
    return source_filter
