"""rag_dpo_amd — MI355X-native dense retrieval hot path for RAG-DPO (embed -> L2-normalise -> cosine top-k).

Only the hot path of SURVEY.md §8 lives here: HIP kernels + C-ABI (csrc/, include/rdx.h) and the host-side
mirror of the two objects the reference's retriever consumes (`collection`, `embedding_provider`).
"""
__version__ = "0.1.0"
