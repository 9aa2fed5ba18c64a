"""CPU: pins the C oracle with known answers and an independent numpy float64 restatement.
(Parity of the top-k arithmetic with chromadb==1.4.1 itself is UNPINNED: see oracle/rdx_oracle.c.)"""
import numpy as np

from rag_dpo_amd import synth


def test_normalize_matches_numpy(oracle):
    x = synth.make_corpus(2000, 1024)
    a, b = oracle.normalize_rows(x), oracle.normalize_numpy(x)
    assert np.abs(a.astype(np.float64) - b).max() <= 2 ** -24      # at most one fp32 ulp of a unit vector entry
    assert abs(np.linalg.norm(a.astype(np.float64), axis=1) - 1).max() < 1e-6
    z = np.zeros((1, 1024), np.float32)
    assert (oracle.normalize_rows(z) == 0).all()               # reference: x / max(|x|, 1e-12)


def test_topk_matches_numpy_float64(oracle):
    corpus = oracle.normalize_rows(synth.make_corpus(6000, 1024))
    q = synth.make_queries(24, 1024, corpus)
    for k in (1, 10, 100):
        s, r, c = oracle.cosine_topk(corpus, q, k)
        s2, r2, c2 = oracle.topk_numpy(corpus, q, k)
        assert (c == k).all() and (c2 == k).all()
        assert np.abs(s - s2).max() <= 2 ** -23
        assert (r == r2).mean() > 0.999          # only sub-ulp near-ties may swap between summation orders
        assert (np.diff(s.astype(np.float64), axis=1) <= 0).all()


def test_known_answers(oracle):
    d = 1024
    e0 = np.zeros(d, np.float32); e0[0] = 1
    e1 = np.zeros(d, np.float32); e1[1] = 1
    corpus = oracle.normalize_rows(np.stack([e1, 3 * e0, -e0, e0, e0 + e1]))
    s, r, c = oracle.cosine_topk(corpus, e0[None], 5)
    assert r[0].tolist() == [1, 3, 4, 0, 2]
    dist = 1.0 - s[0].astype(np.float64)
    assert abs(dist[0]) < 1e-7 and abs(dist[3] - 1) < 1e-7 and abs(dist[4] - 2) < 1e-7
    s, r, c = oracle.cosine_topk(corpus, e0[None], 8)       # n_results > N -> short result
    assert c[0] == 5 and (r[0, 5:] == -1).all() and np.isneginf(s[0, 5:]).all()


def test_mask_is_prefilter(oracle):
    corpus = oracle.normalize_rows(synth.make_corpus(3000, 256))
    q = synth.make_queries(8, 256)
    allow = np.random.default_rng(0).random(3000) < 0.02
    s, r, c = oracle.cosine_topk(corpus, q, 10, allow)
    sub = np.flatnonzero(allow)
    s2, r2, c2 = oracle.cosine_topk(corpus[sub], q, 10)
    np.testing.assert_array_equal(r, sub[r2])                  # filter-then-knn == knn on the subset
    np.testing.assert_array_equal(s, s2)
    assert (c == 10).all()


def test_merge(oracle):
    corpus = oracle.normalize_rows(synth.make_corpus(4000, 128))
    q = synth.make_queries(9, 128, corpus)
    k = 7
    full = oracle.cosine_topk(corpus, q, k)
    parts = [oracle.cosine_topk(corpus[a:b], q, k) for a, b in ((0, 1500), (1500, 1501), (1501, 4000))]
    ps = np.stack([p[0] for p in parts]); pc = np.stack([p[2] for p in parts])
    pr = np.stack([np.where(p[1] >= 0, p[1] + off, -1) for p, off in zip(parts, (0, 1500, 1501))])
    s, r, c = oracle.merge_topk(ps, pr, pc, k)
    np.testing.assert_array_equal(r, full[1]); np.testing.assert_array_equal(s, full[0])
