"""The oracle's SCORE pinned to the reference's own arithmetic: tests/golden/cosine_golden.{json,npy} were captured by running
`TopicMatcher.similarity` of the imported reference (/root/reference/src/utils/rgpd_topics.py:167-177, float(np.dot(vec_a, vec_b)) on
provider embeddings; tests/golden/make_cosine_golden.py) on seeded d = 1024 unit vectors — random, duplicate, orthogonal,
antipodal, near-duplicate and badly scaled pairs. CPU: the C oracle within 1e-6 of the fixture. GPU: rdx_search within
north_star's 1e-4 (in fact 1e-6) and in the order the reference's similarities induce. Top-k order on exact ties, the tie rule
and `distance = 1 - cos` remain chromadb's contract (absent offline): parity there stays unpinned."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TOL_ORACLE = 1e-6     # fp32 rounding of a fp64-accumulated dot of unit vectors: <= 6e-8, plus one more ulp per re-normalisation
TOL_NORTH_STAR = 1e-4  # BASELINE.json: "scores within 1e-4 fp32"


def load():
    g = json.load(open(os.path.join(HERE, "golden", "cosine_golden.json")))
    v = np.load(os.path.join(HERE, "golden", "cosine_golden.npy"), allow_pickle=False)
    assert v.dtype == np.float32 and v.shape == (len(g["names"]), g["dim"])
    return g, v


def test_fixture_is_what_it_says():
    g, v = load()
    assert len(g["pairs"]) >= 20 and {p["kind"].split(" ")[0] for p in g["pairs"]} >= {"random", "duplicate", "orthogonal", "antipodal", "near-duplicate"}
    assert np.allclose(np.linalg.norm(v.astype(np.float64), axis=1), 1.0, atol=1e-6)      # provider output: unit vectors
    for p in g["pairs"]:   # the reference's expression, restated: float64 dot of the fp32 values
        assert p["similarity"] == float(np.dot(v[p["ia"]].astype(np.float64), v[p["ib"]].astype(np.float64)))


def test_oracle_scores_match_the_reference_cosine(oracle):
    g, v = load()
    for p in g["pairs"]:
        s = oracle.scores(v[p["ib"]][None, :], v[p["ia"]])[0]                     # stored vectors as they are
        assert s.dtype == np.float32 and abs(float(s) - p["similarity"]) <= TOL_ORACLE, p
        # through the whole oracle search (normalises rows and query once more, as ingest and K1 do)
        es, er, ec = oracle.cosine_topk(oracle.normalize_rows(v), v[p["ia"]][None, :], len(v))
        got = float(es[0][list(er[0]).index(p["ib"])])
        assert abs(got - p["similarity"]) <= TOL_ORACLE, p
    for rk in g["rankings"]:
        es, er, ec = oracle.cosine_topk(oracle.normalize_rows(v), v[rk["iq"]][None, :], len(v))
        _check_order(rk, es[0], er[0])


def _check_order(rk, scores, rows):
    ref = np.asarray(rk["similarities"])
    assert sorted(rows.tolist()) == list(range(len(ref)))
    assert np.abs(scores.astype(np.float64) - ref[rows]).max() <= TOL_ORACLE
    # the returned order is the reference's order wherever the reference separates two rows by more than the tolerance
    for a, b in zip(rows[:-1], rows[1:]):
        assert ref[a] >= ref[b] - 2 * TOL_ORACLE, (rk["q"], a, b)


@pytest.mark.gpu
def test_hip_scores_match_the_reference_cosine():
    from rag_dpo_amd.engine import HipIndex
    g, v = load()
    n = len(v)
    for opt in ("force_exact", "force_fast"):                  # the exact scan (the reference's own corpus size) and the MFMA path
        ix = HipIndex(g["dim"])
        ix.add(v)
        ix.set_option(opt, 1)
        qs = sorted({p["ia"] for p in g["pairs"]})
        s, r, c = ix.search(v[qs], n)
        assert (c == n).all()
        for p in g["pairs"]:
            b = qs.index(p["ia"])
            got = float(s[b][list(r[b]).index(p["ib"])])
            assert abs(got - p["similarity"]) <= TOL_NORTH_STAR, (opt, p, got)
            assert abs(got - p["similarity"]) <= TOL_ORACLE, (opt, p, got)      # (what the fixed-order fp64 sum actually delivers)
        for rk in g["rankings"]:
            s1, r1, c1 = ix.search(v[rk["iq"]][None, :], n)
            _check_order(rk, s1[0], r1[0])
        ix.close()
