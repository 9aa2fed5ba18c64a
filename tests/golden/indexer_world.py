"""Synthetic project tree + chunks shared by make_indexer_golden.py (capture: runs the IMPORTED reference indexer in the
build container) and tests/test_indexer_golden.py (replay: this repo's indexer only). The chunk records follow the
schema process_and_chunk.py writes to processed_chunks.jsonl (reference src/processing/create_chromadb_index.py:315-360)."""
import json
import os

import fixture_world as W

BATCH = 5
POISON = "ce texte fait échouer l'embedder"

HELPER_PATHS = [
    "data/raw/cnil/html/guide_aipd.html", "data\\raw\\cnil\\pdf\\deliberation_2023.PDF", "data/raw/Entreprise/politique_rh.docx",
    "data/raw/cnil/docs/registre_modele.xlsx", "data/custom/notes.odt", "data/raw/cnil/docs/archive.zip",
    "internal/policy.htm", "data/raw/cnil/docs/Modèle_clause.doc", "data/raw/cnil/docs/template.ods", "",
]

NATURES = ["GUIDE", "DOCTRINE", "SANCTION", "TECHNIQUE"]


def base_chunks():
    out = []
    for i in range(23):
        path = HELPER_PATHS[i % 9]
        c = {"chunk_id": f"doc{i % 7}_chunk_{i:03d}", "document_id": f"doc{i % 7}", "document_path": path,
             "text": f"Texte du chunk numéro {i} : durée de conservation, base légale, AIPD.  Espaces   multiples {i}",
             "chunk_nature": NATURES[i % 4], "document_nature": NATURES[(i + 1) % 4], "chunk_index": i % 6,
             "confidence": [0.91, 0.5, 0.73][i % 3], "method": "llm" if i % 2 else "heuristic"}
        if i % 3 != 0:
            c["heading"] = f"Section {i}" if i != 4 else "T" * 260           # one heading beyond the 200-char clip
        if i % 4 == 0:
            c["sectors"] = ["santé", "rh"] if i % 8 == 0 else []
        if i % 5 == 0:
            c["title"] = ("Titre du document " + str(i)) * (30 if i == 10 else 1)   # one title beyond the 300-char clip
        if i % 6 == 1:
            c["source_url"] = f"https://www.cnil.fr/fr/page-{i}"
        if i % 6 == 2:
            c["parent_url"] = f"https://www.cnil.fr/fr/parent-{i}"
        if i % 7 == 3:
            c["page_info"] = f"Pages {i}-{i + 2}"
        if i % 9 == 5:
            c["file_type"] = "archive"
        if i == 12:
            c["text"] = POISON                                               # its whole batch fails in the embedder
        if i == 17:
            del c["chunk_id"]                                                # default id chunk_{batch start}
        if i in (19, 21):
            for k in ("chunk_nature", "document_nature", "chunk_index", "confidence", "method", "document_id"):
                c.pop(k, None)                                               # every default of the metadata dict
        out.append(c)
    out.append(dict(out[20], text="même id dans le même lot"))               # duplicate id inside one batch -> add() raises
    return out


def extra_chunks():
    """second ('update' mode) run: two ids that exist already + three new ones"""
    b = base_chunks()
    new = [{"chunk_id": f"new_chunk_{j}", "document_path": "data/raw/entreprise/charte.pdf", "text": f"nouveau {j}",
            "heading": "Charte", "chunk_nature": "GUIDE"} for j in range(3)]
    return [b[0], b[1]] + new


def write_project(root):
    cnil = os.path.join(root, "data", "raw", "cnil")
    os.makedirs(cnil)
    os.makedirs(os.path.join(root, "data", "metadata"))
    with open(os.path.join(cnil, "processed_chunks.jsonl"), "w", encoding="utf-8") as f:
        for i, c in enumerate(base_chunks()):
            f.write(json.dumps(c, ensure_ascii=False) + "\n")
            if i == 6:
                f.write("{this line is not json\n")
    keep = {"html": [{"metadata": {"file_path": "data/raw/cnil/html/guide_aipd.html", "url": "https://www.cnil.fr/fr/guide-aipd"}},
                     {"url": "https://www.cnil.fr/fr/top-level", "metadata": {"file_path": "internal/policy.htm"}},
                     {"metadata": {"url": "https://www.cnil.fr/fr/sans-chemin"}}],
            "pdfs": [{"parent_url": "https://www.cnil.fr/fr/deliberations", "metadata": {"file_path": "data\\raw\\cnil\\pdf\\deliberation_2023.PDF"}}],
            "docs": [{"metadata": {"file_path": "data/raw/cnil/docs/registre_modele.xlsx", "source_url": "https://www.cnil.fr/fr/registre"}},
                     {"metadata": {"file_path": "data/raw/cnil/docs/template.ods"}}],
            "pdf": [{"metadata": {"file_path": "data/raw/cnil/docs/archive.zip", "url": "https://ignored.example"}}]}
    with open(os.path.join(cnil, "keep_manifest.json"), "w", encoding="utf-8") as f:
        json.dump(keep, f, ensure_ascii=False)
    return os.path.join(cnil, "processed_chunks.jsonl"), os.path.join(cnil, "keep_manifest.json")


class FlakyEmbedder(W.HashEmbedder):
    """HashEmbedder that raises when a batch holds the poisoned text (the reference counts the batch as errors)"""

    def embed(self, texts):
        if any(POISON in t for t in texts):
            raise RuntimeError("embedder down")
        return super().embed(texts)
