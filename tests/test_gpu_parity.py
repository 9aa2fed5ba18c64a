"""GPU parity: the HIP path (through the C-ABI, rag_dpo_amd/engine.py) against the CPU oracle on the same
seeded inputs. Bar: row ids bit-exact, scores bit-equal (both sides use the fixed fp64 lane-order sum)."""
import numpy as np
import pytest

from rag_dpo_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from rag_dpo_amd import engine
    return engine


def _index(eng, corpus, **opts):
    ix = eng.HipIndex(corpus.shape[1])
    ix.add(corpus)
    for k, v in opts.items():
        ix.set_option(k, v)
    return ix


def _check(oracle, ix, corpus, q, k, allow=None, expect_path=None):
    ch = oracle.normalize_rows(corpus)
    es, er, ec = oracle.cosine_topk(ch, q, k, allow)
    bits = oracle.pack_mask(allow, corpus.shape[0])
    gs, gr, gc = ix.search(q, k, bits)
    st = ix.last_stats()
    np.testing.assert_array_equal(gc, ec)
    np.testing.assert_array_equal(gr, er)
    np.testing.assert_array_equal(gs, es)
    if expect_path is not None:
        assert st["path"] == expect_path, st
    return st


def test_normalize_bit_exact(eng, oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((777, 1024)).astype(np.float32)
    x[3] = 0.0                      # zero vector -> stays zero (reference clamps the norm at 1e-12)
    x[4] *= 1e-20
    x[5] *= 1e15
    x[6, 1:] = 0.0
    np.testing.assert_array_equal(eng.l2_normalize(x), oracle.normalize_rows(x))
    ix = _index(eng, x)
    np.testing.assert_array_equal(ix.get(np.arange(777)), oracle.normalize_rows(x))
    for d in (64, 384, 768, 4096):
        y = rng.standard_normal((33, d)).astype(np.float32)
        np.testing.assert_array_equal(eng.l2_normalize(y), oracle.normalize_rows(y))


def test_exact_path_small(eng, oracle):
    corpus = synth.make_corpus(3000, 1024)
    q = synth.make_queries(5, 1024, corpus)
    ix = _index(eng, corpus)
    st = _check(oracle, ix, corpus, q, 10, expect_path=1)
    assert st["exact_queries"] == 5
    _check(oracle, ix, corpus, q[:1], 50, expect_path=1)   # the reference's production shape: B=1, k=50


@pytest.mark.parametrize("d", [64, 200, 320, 700, 1024, 1280, 2048])
def test_exact_scores_and_rescore_every_width(eng, oracle, d):
    """K5a keeps the queries in registers for dim <= 1024 (one to four float4 groups per lane, the last one partly empty when dim is
    not a multiple of 256) and falls back to the LDS form above; K4's re-score has the same split. Both paths, six queries (a full
    and a partial group of four), with and without a row bitmap."""
    corpus = synth.make_corpus(3000, d)
    q = synth.make_queries(6, d, corpus)
    ix = _index(eng, corpus)
    st = _check(oracle, ix, corpus, q, 20, expect_path=1)
    assert st["exact_queries"] == 6
    allow = np.random.default_rng(d).random(corpus.shape[0]) < 0.3
    _check(oracle, ix, corpus, q, 20, allow, expect_path=1)
    ix.close()
    corpus = synth.make_corpus(30_000, d)
    q = synth.make_queries(70, d, corpus)
    ix = _index(eng, corpus, force_fast=1)
    _check(oracle, ix, corpus, q, 30, expect_path=0)
    ix.close()


@pytest.mark.parametrize("n,b,k", [(20000, 64, 10), (16919, 4, 50), (30011, 130, 10), (25000, 257, 100)])
def test_fast_path_parity(eng, oracle, n, b, k):
    corpus = synth.make_corpus(n, 1024)
    q = synth.make_queries(b, 1024, corpus)
    ix = _index(eng, corpus, force_fast=1)
    st = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert st["exact_queries"] == 0, st
    assert st["emitted"] >= b * k


def test_baseline_config2(eng, oracle):
    """BASELINE.json config 2 as written: 100k x 1024 fp32, batch 64, top-10 — the whole corpus against the oracle
    (ids and score bits), default path selection (MFMA scan, LDS-resident 64-query tile), §8(d) inputs"""
    corpus = synth.make_corpus(100_000, 1024)
    q = synth.make_queries(64, 1024, corpus)
    ix = _index(eng, corpus)
    st = _check(oracle, ix, corpus, q, 10, expect_path=0)
    assert st["exact_queries"] == 0, st
    ix.close()


def test_search_in_two_halves(eng, oracle):
    """rdx_search_async + rdx_search_wait (the form the multi-GPU exchange uses): same results as rdx_search; when candidate
    segments overflow the wait runs the fallback passes and reports it; a following call completes a pending search itself"""
    import torch
    corpus = synth.make_corpus(40_000, 1024)
    q = synth.make_queries(130, 1024, corpus)
    es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q, 50)
    ix = _index(eng, corpus, force_fast=1)
    qd = torch.from_numpy(q).cuda()
    s = torch.empty((130, 50), dtype=torch.float32, device="cuda"); r = torch.empty((130, 50), dtype=torch.int64, device="cuda")
    c = torch.empty((130,), dtype=torch.int32, device="cuda")
    f = torch.full((4,), 7, dtype=torch.int32, device="cuda")

    def check():
        torch.cuda.synchronize()
        np.testing.assert_array_equal(r.cpu().numpy(), er); np.testing.assert_array_equal(s.cpu().numpy(), es)
        np.testing.assert_array_equal(c.cpu().numpy(), ec)
    ix.search_device_async(qd, 50, s, r, c)
    assert ix.search_wait() is False and ix.last_stats()["exact_queries"] == 0
    check()
    assert ix.search_wait() is False                       # nothing pending: a no-op
    ix.set_option("cand_cap", 8)                           # 8 slots per segment cannot hold k = 50: overflow -> fallback passes
    s.zero_(); r.zero_(); c.zero_()
    ix.search_device_async(qd, 50, s, r, c)
    assert ix.search_wait() is True
    check()
    # the flags word (include/rdx.h): 0 behind a complete first pass; 1 while the partial is incomplete — what a consumer enqueued
    # behind the search (the all-gather) sees — and 0 again once the wait has run the fallback passes
    ix.set_option("cand_cap", 0)
    ix.search_device_async(qd, 50, s, r, c, f)
    seen = f.clone()                                       # stream-ordered behind the search's last kernel, before the host half
    assert ix.search_wait() is False
    assert seen.tolist() == [0, 0, 0, 0] and f.tolist() == [0, 0, 0, 0]
    ix.set_option("cand_cap", 8)
    s.zero_(); r.zero_(); c.zero_()
    q2 = qd.clone()
    ix.search_device_async(q2, 50, s, r, c, f)
    seen = f.clone()
    q2.fill_(float("nan"))                                 # `queries` belongs to the caller again once the search is enqueued:
    assert ix.search_wait() is True                        #   the fallback passes work from the index's own normalised copy
    assert seen.tolist() == [1, 0, 0, 0] and f.tolist() == [0, 0, 0, 0]
    check()
    ix.set_option("retry", 0)                              # straight to the exact scan of the overflowed queries: same contract
    s.zero_(); r.zero_(); c.zero_()
    q2 = qd.clone()
    ix.search_device_async(q2, 50, s, r, c, f)
    seen = f.clone()
    q2.fill_(float("nan"))
    assert ix.search_wait() is True
    assert seen.tolist() == [1, 0, 0, 0] and f.tolist() == [0, 0, 0, 0]
    check()
    ix.set_option("retry", 1)
    s.zero_(); r.zero_(); c.zero_()
    ix.search_device_async(qd, 50, s, r, c)                # never waited for: the next call on the index completes it first
    gs, gr, gc = ix.search(q[:7], 50)
    np.testing.assert_array_equal(gr, er[:7])
    check()
    ix.close()


@pytest.mark.parametrize("b", [129, 200, 256])
def test_half_tile_bootstrap(eng, oracle, b):
    """option half_boot (default 1): 129..256 queries sample their threshold with the 128-query kernel as two query tiles per sampled
    corpus tile; same answers as with the one-tile bootstrap, with a bitmap too"""
    corpus = synth.make_corpus(90_000, 256)
    q = synth.make_queries(b, 256, corpus)
    allow = np.random.default_rng(b).random(corpus.shape[0]) < 0.5
    for hb in (1, 0):
        ix = _index(eng, corpus, half_boot=hb)
        st = _check(oracle, ix, corpus, q, 20, expect_path=0)
        assert st["exact_queries"] == 0 and st["retried_queries"] == 0, (hb, st)
        _check(oracle, ix, corpus, q, 20, allow, expect_path=0)
        ix.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_fused_epilogue_variant(eng, oracle, fused):
    """option fuse_epilogue (default 1): the B > 128 main scan whose emit check rides in the next tile's first k-step — several
    tiles per stream (so that the fused step runs), a ragged last tile, a row bitmap, planted near-duplicates, against the
    oracle; and the same with the option off (the stand-alone per-tile check, which odd k-step counts always take)"""
    corpus = synth.make_corpus(140_001, 256)           # 547 tiles over 64 streams: 8-9 tiles per stream; 4 k-steps per tile
    q = synth.make_queries(300, 256, corpus)
    ix = _index(eng, corpus, force_fast=1, fuse_epilogue=fused)
    st = _check(oracle, ix, corpus, q, 10, expect_path=0)
    assert st["exact_queries"] == 0
    allow = np.random.default_rng(1).random(corpus.shape[0]) < 0.4
    _check(oracle, ix, corpus, q, 20, allow, expect_path=0)
    ix.close()


@pytest.mark.parametrize("rows,dim,b", [(140_001, 256, 600), (70_000, 1024, 1000)])
def test_one_wave_per_simd_layout(eng, oracle, rows, dim, b):
    """developer option wave_layout = 1 (csrc/scan_w4.hpp): the B > 128 fused main scan with 4 waves per workgroup, each owning 64
    rows x 256 queries (256 pinned accumulators, inline-asm MFMAs, the rare emit path staged through LDS) — same candidates, so the
    same answers: several tiles per stream, ragged last tiles (225 and 112 rows: a wave with 33 rows, waves with none), three and
    four query tiles (the last one partly filled; launches of at most two tiles take 128-query workgroups), a row bitmap, against
    the oracle; the candidate count equals the shipped kernel's"""
    corpus = synth.make_corpus(rows, dim)
    q = synth.make_queries(b, dim, corpus)
    ix = _index(eng, corpus, force_fast=1, wave_layout=1)
    st = _check(oracle, ix, corpus, q, 10, expect_path=0)
    assert st["exact_queries"] == 0
    allow = np.random.default_rng(1).random(corpus.shape[0]) < 0.4
    _check(oracle, ix, corpus, q, 20, allow, expect_path=0)
    e1 = _check(oracle, ix, corpus, q, 10, expect_path=0)["emitted"]
    ix.set_option("wave_layout", 0)
    assert _check(oracle, ix, corpus, q, 10, expect_path=0)["emitted"] == e1
    ix.close()


def test_xcd_shares_ignore_idle_workgroups(eng):
    """nq = 600 -> three query tiles: 32 workgroups per XCD = 10 streams x 3 + 2 idle ones, which return before they stamp
    their times. The XCD re-weighting must skip them (their stamp slots are stale memory): shares stay near an eighth and
    the finish spread stays a sane number over repeated searches."""
    rng = np.random.default_rng(3)
    n, d = 270_000, 128
    ix = eng.HipIndex(d)
    for a in range(0, n, 90_000):
        ix.add(rng.standard_normal((90_000, d)).astype(np.float32))
    q = rng.standard_normal((600, d)).astype(np.float32)
    for _ in range(5):
        s, r, c = ix.search(q, 10)
        st = ix.last_stats()
        assert st["path"] == 0
        assert 0.85 <= st["xcd_share_min"] <= 1.0 <= st["xcd_share_max"] <= 1.18, st
        assert 0.0 <= st["xcd_finish_spread_ms"] < 1.0, st
    ix.set_option("force_exact", 1)
    xs, xr, xc = ix.search(q[:8], 10)
    assert (xr == r[:8]).all() and (xs == s[:8]).all()
    ix.close()


def test_fast_path_other_dims(eng, oracle):
    rng = np.random.default_rng(11)
    for d in (768, 200, 64):
        corpus = rng.standard_normal((9000, d)).astype(np.float32)
        q = rng.standard_normal((70, d)).astype(np.float32)
        ix = _index(eng, corpus, force_fast=1)
        _check(oracle, ix, corpus, q, 10, expect_path=0)


def test_where_mask(eng, oracle):
    corpus = synth.make_corpus(20000, 1024)
    q = synth.make_queries(40, 1024, corpus)
    rng = np.random.default_rng(3)
    for frac in (0.5, 0.01, 0.0003):
        allow = rng.random(20000) < frac
        for opts in ({"force_fast": 1}, {"force_exact": 1}):
            ix = _index(eng, corpus, **opts)
            _check(oracle, ix, corpus, q, 10, allow)
    none = np.zeros(20000, dtype=bool)
    ix = _index(eng, corpus, force_fast=1)
    s, r, c = ix.search(q, 10, oracle.pack_mask(none, 20000))
    assert (c == 0).all() and (r == -1).all() and np.isneginf(s).all()


def test_k_larger_than_corpus_and_tiny(eng, oracle):
    corpus = synth.make_corpus(7, 1024, duplicates=False)
    q = synth.make_queries(3, 1024)
    ix = _index(eng, corpus)
    s, r, c = ix.search(q, 10)
    assert (c == 7).all() and (r[:, 7:] == -1).all()
    _check(oracle, ix, corpus, q, 10)
    _check(oracle, ix, corpus, q, 1)
    empty = eng.HipIndex(1024)
    s, r, c = empty.search(q, 5)
    assert (c == 0).all() and (r == -1).all()


def test_known_answers(eng):
    d = 1024
    e0 = np.zeros(d, np.float32); e0[0] = 1
    e1 = np.zeros(d, np.float32); e1[1] = 1
    corpus = np.stack([e1, 3 * e0, -e0, e0, e0 + e1])
    ix = _index(eng, corpus)
    s, r, c = ix.search(e0[None], 5)
    assert r[0].tolist() == [1, 3, 4, 0, 2]            # duplicates: lower row id first
    dist = 1.0 - s[0].astype(np.float64)
    assert abs(dist[0]) < 1e-6 and abs(dist[1]) < 1e-6  # identical -> 0
    assert abs(dist[3] - 1) < 1e-6                      # orthogonal -> 1
    assert abs(dist[4] - 2) < 1e-6                      # antipodal -> 2
    assert abs(dist[2] - (1 - 2 ** -0.5)) < 1e-6


def test_all_rows_identical_ties(eng, oracle):
    corpus = np.tile(np.random.default_rng(1).standard_normal((1, 1024)).astype(np.float32), (5000, 1))
    q = synth.make_queries(66, 1024)
    ix = _index(eng, corpus, force_fast=1)
    st = _check(oracle, ix, corpus, q, 10)
    assert st["exact_queries"] == 66        # every query overflows its candidate list -> exact scan


def test_candidate_overflow_falls_back(eng, oracle):
    corpus = synth.make_corpus(20000, 1024)
    q = synth.make_queries(64, 1024, corpus)
    ix = _index(eng, corpus, force_fast=1, cand_cap=1)      # one slot per (query, stream) segment cannot hold k=50's hits
    st = _check(oracle, ix, corpus, q, 50, expect_path=0)
    assert st["retried_queries"] > 0                       # overflowed queries get the second MFMA pass (own, large segments)
    ix.set_option("retry", 0)
    st = _check(oracle, ix, corpus, q, 50, expect_path=0)
    assert st["exact_queries"] > 0 and st["retried_queries"] == 0   # ... or the exact full scan when that is switched off


def test_update_get_compact(eng, oracle):
    corpus = synth.make_corpus(6000, 1024)
    q = synth.make_queries(20, 1024, corpus)
    ix = _index(eng, corpus[:3000])
    ix.add(corpus[3000:])
    assert len(ix) == 6000
    new = np.random.default_rng(8).standard_normal((3, 1024)).astype(np.float32)
    ix.update([5, 4000, 5999], new)
    ref = corpus.copy(); ref[[5, 4000, 5999]] = new
    for opts in ({"force_fast": 1}, {"force_exact": 1}):
        for k, v in opts.items():
            ix.set_option(k, v)
        _check(oracle, ix, ref, q, 10)
        ix.set_option("force_fast", 0); ix.set_option("force_exact", 0)
    keep = np.setdiff1d(np.arange(6000), np.arange(100, 6000, 7))
    ix.compact(keep)
    assert len(ix) == keep.size
    ix.set_option("force_fast", 1)
    _check(oracle, ix, ref[keep], q, 10)
    np.testing.assert_array_equal(ix.get([0, 1, keep.size - 1]), oracle.normalize_rows(ref[keep][[0, 1, -1]]))


def test_bf16_corpus(eng, oracle):
    import torch
    corpus = synth.make_corpus(12000, 1024)
    cb = torch.from_numpy(corpus).to(torch.bfloat16)
    ix = eng.HipIndex(1024)
    ix.add_bf16(cb)
    ix.set_option("force_fast", 1)
    q = synth.make_queries(70, 1024, corpus)
    _check(oracle, ix, cb.to(torch.float32).numpy(), q, 10, expect_path=0)


def test_compact_master_for_bf16_corpora(eng, oracle):
    """option compact_master (BASELINE config 5: a bf16 corpus at 2 + 2 instead of 4 + 2 B/element): the exact copy holds the raw
    bf16 rows + one divisor per row and every normalised element is recomputed as (float)((double)x / den) where it is
    needed. Ids, score bits and the rows get() returns must equal the oracle on the bf16-widened values and the default
    (fp32 master) index — on the MFMA path, the exact path, through a row bitmap, after growth and after compaction."""
    import torch
    rng = np.random.default_rng(3)
    corpus = synth.make_corpus(30000, 1024)
    cb = torch.from_numpy(corpus).to(torch.bfloat16)
    wide = cb.to(torch.float32).numpy()
    ix, ref = eng.HipIndex(1024), eng.HipIndex(1024)
    ix.set_option("compact_master", 1)
    for a in range(0, 30000, 7000):                   # several adds: the raw rows and divisors survive growth
        ix.add_bf16(cb[a:a + 7000]); ref.add_bf16(cb[a:a + 7000])
    q = synth.make_queries(70, 1024, corpus)
    np.testing.assert_array_equal(ix.get(np.arange(0, 30000, 7)), oracle.normalize_rows(wide)[::7])
    for opt in ("force_fast", "force_exact"):
        ix.set_option(opt, 1); ref.set_option(opt, 1)
        _check(oracle, ix, wide, q, 10, expect_path=0 if opt == "force_fast" else 1)
        allow = rng.random(30000) < 0.3
        _check(oracle, ix, wide, q[:9], 50, allow)
        a, b = ix.search(q, 100), ref.search(q, 100)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        ix.set_option(opt, 0); ref.set_option(opt, 0)
    keep = np.flatnonzero(rng.random(30000) < 0.5)
    ix.compact(keep)
    ix.set_option("force_fast", 1)
    _check(oracle, ix, wide[keep], q, 10, expect_path=0)
    with pytest.raises(Exception, match="compact"):
        ix.add(corpus[:5])                             # fp32 rows cannot enter a compact (bf16) master
    with pytest.raises(Exception, match="compact"):
        ix.update(np.array([0]), corpus[:1])
    with pytest.raises(Exception, match="empty"):
        ref.set_option("compact_master", 1)            # only while the index is empty
    ix.close(); ref.close()


def test_rejects_nan(eng):
    corpus = synth.make_corpus(100, 1024)
    bad = corpus.copy(); bad[50, 7] = np.nan
    ix = eng.HipIndex(1024)
    with pytest.raises(ValueError):
        ix.add(bad)
    assert len(ix) == 0
    ix.add(corpus)
    q = synth.make_queries(2, 1024); q[1, 0] = np.inf
    with pytest.raises(ValueError):
        ix.search(q, 3)
    with pytest.raises(ValueError):
        ix.search(np.zeros((2, 512), np.float32), 3)


def test_merge_parity(eng, oracle):
    rng = np.random.default_rng(2)
    P, B, k = 8, 50, 10
    ps = np.sort(rng.standard_normal((P, B, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ps[3] = ps[2]                                   # equal scores across parts -> tie by row id
    pr = rng.permutation(P * B * k).reshape(P, B, k).astype(np.int64)
    pc = rng.integers(0, k + 1, size=(P, B)).astype(np.int32)
    es, er, ec = oracle.merge_topk(ps, pr, pc, k)
    gs, gr, gc = eng.merge_topk(ps, pr, pc, k)
    np.testing.assert_array_equal(gc, ec)
    for b in range(B):
        np.testing.assert_array_equal(gr[b, :gc[b]], er[b, :ec[b]])
        np.testing.assert_array_equal(gs[b, :gc[b]], es[b, :ec[b]])


@pytest.mark.parametrize("P,k", [(2, 2500), (3, 4096), (8, 700), (64, 65)])
def test_merge_more_candidates_than_one_launch(eng, oracle, P, k):
    """n_parts * k > 4096: rdx_merge_topk folds the parts pairwise (k_merge_pair) — same lists as the oracle's merge"""
    rng = np.random.default_rng(P * 10000 + k)
    B = 6
    ps = np.sort(rng.standard_normal((P, B, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ps[-1] = ps[0]                                  # equal scores across parts -> tie by row id
    pr = np.empty((P, B, k), dtype=np.int64)
    for b in range(B):
        rows = rng.permutation(P * k).reshape(P, k)
        for p in range(P):
            pr[p, b] = rows[p]                      # no row in two parts
    # every list sorted (score desc, row asc), as a shard returns it
    for p in range(P):
        for b in range(B):
            o = np.lexsort((pr[p, b], -ps[p, b].astype(np.float64)))
            ps[p, b], pr[p, b] = ps[p, b][o], pr[p, b][o]
    pc = rng.integers(0, k + 1, size=(P, B)).astype(np.int32)
    pc[0, 0] = k; pc[1, 0] = k; pc[0, 1] = 0
    es, er, ec = oracle.merge_topk(ps, pr, pc, k)
    gs, gr, gc = eng.merge_topk(ps, pr, pc, k)
    np.testing.assert_array_equal(gc, ec)
    for b in range(B):
        np.testing.assert_array_equal(gr[b, :gc[b]], er[b, :ec[b]])
        np.testing.assert_array_equal(gs[b, :gc[b]], es[b, :ec[b]])
        assert (gr[b, gc[b]:] == -1).all()


def test_packed_merge_and_signal(eng, oracle):
    """rdx_merge_topk_packed over the all-gather layout (rows | scores | counts | flags per part) + rdx_signal: the merged lists
    equal the oracle's merge, and the host reads the OR of the parts' flags words from the signal (same answer wherever it runs)"""
    import ctypes
    import torch
    from rag_dpo_amd import _lib as L
    from rag_dpo_amd.sharded import ShardedSearcher, packed_bytes
    lib = L.load()
    rng = np.random.default_rng(12)
    P, B, k = 5, 33, 7
    ps = np.sort(rng.standard_normal((P, B, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    pr = rng.permutation(P * B * k).reshape(P, B, k).astype(np.int64)
    pc = rng.integers(0, k + 1, size=(P, B)).astype(np.int32)
    es, er, ec = oracle.merge_topk(ps, pr, pc, k)
    stride = (packed_bytes(B, k) + 15) // 16 * 16
    buf = torch.zeros(P * stride, dtype=torch.uint8, device="cuda")
    parts = [ShardedSearcher.views(buf[p * stride:(p + 1) * stride], B, k) for p in range(P)]
    for p in range(P):
        parts[p][0].copy_(torch.from_numpy(ps[p])); parts[p][1].copy_(torch.from_numpy(pr[p])); parts[p][2].copy_(torch.from_numpy(pc[p]))
    os_ = torch.empty((B, k), dtype=torch.float32, device="cuda"); or_ = torch.empty((B, k), dtype=torch.int64, device="cuda")
    oc = torch.empty((B,), dtype=torch.int32, device="cuda")
    sig = ctypes.c_void_p()
    L.check(lib.rdx_signal_create(0, ctypes.byref(sig)))
    v = ctypes.c_int32(-1)
    with pytest.raises(L.RdxError):
        L.check(lib.rdx_signal_wait(sig, None, ctypes.byref(v)))          # no merge was given the signal yet
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for flagged in ([], [3], [0, 4], []):
        for p in range(P):
            parts[p][3][0] = 1 if p in flagged else 0
        L.check(lib.rdx_merge_topk_packed(0, ptr(buf), stride, P, B, k, ptr(os_), ptr(or_), ptr(oc), sig, stream))
        L.check(lib.rdx_signal_wait(sig, stream, ctypes.byref(v)))
        assert v.value == (1 if flagged else 0)
        torch.cuda.synchronize()
        gc = oc.cpu().numpy()
        np.testing.assert_array_equal(gc, ec)
        for b in range(B):
            np.testing.assert_array_equal(or_.cpu().numpy()[b, :gc[b]], er[b, :ec[b]])
            np.testing.assert_array_equal(os_.cpu().numpy()[b, :gc[b]], es[b, :ec[b]])
    with pytest.raises(ValueError):                                       # a stride that does not cover the flags word
        L.check(lib.rdx_merge_topk_packed(0, ptr(buf), (B * k * 12 + B * 4 + 15) // 16 * 16 - 16, P, B, k, ptr(os_), ptr(or_), ptr(oc), None, stream))
    L.check(lib.rdx_signal_destroy(sig))


def test_sharded_searcher_world1_flags_path():
    """ShardedSearcher on the product backend with always_exchange (no process group here: the gather of the one part is a
    copy): a search whose segments overflow repeats the exchange once, told so by the merged flags word; a clean one does not"""
    import torch
    from rag_dpo_amd.sharded import HipShard, ShardedSearcher
    from oracle import oracle as O
    corpus = synth.make_corpus(40_000, 1024)
    q = synth.make_queries(130, 1024, corpus)
    es, er, ec = O.cosine_topk(O.normalize_rows(corpus), q, 50)
    sh = HipShard(1024, 0, row_offset=1000)
    sh.add(corpus)
    sh.index.set_option("force_fast", 1)

    ss = ShardedSearcher(sh, always_exchange=True)
    qd = torch.from_numpy(q).cuda()
    for cap, want in ((0, 1), (8, 2), (0, 1)):
        sh.index.set_option("cand_cap", cap)
        n0 = ss.exchanges
        qq = qd.clone()
        ss.search_begin(qq, 50)
        qq.fill_(float("nan"))             # stream-ordered behind the search: allowed (include/rdx.h "Lifetimes")
        s, r, c = ss.search_end()
        torch.cuda.synchronize()
        assert ss.exchanges - n0 == want, (cap, ss.exchanges - n0)
        np.testing.assert_array_equal(r.cpu().numpy(), np.where(er >= 0, er + 1000, -1))
        np.testing.assert_array_equal(s.cpu().numpy(), es); np.testing.assert_array_equal(c.cpu().numpy(), ec)
    sh.close()


@pytest.mark.parametrize("b", [1, 4])
def test_reference_shape_on_the_default_path(eng, oracle, b):
    """BASELINE config 1 at the reference's REAL shape on the path the library picks by itself: 16,919 rows (README.en.md:300-305),
    one query vector per call or the <= 4 reformulations of one question, n_results = 50 (reference src/rag/retriever.py:202,
    215-220, 342) — the exact scan K5 — without and with a `where` bitmap"""
    corpus = synth.make_corpus(16_919, 1024)
    q = synth.make_queries(b, 1024, corpus)
    ix = _index(eng, corpus)
    st = _check(oracle, ix, corpus, q, 50, expect_path=1)
    assert st["exact_queries"] == b
    allow = (np.arange(16_919) % 4 == 1) | (np.random.default_rng(b).random(16_919) < 0.05)   # a chunk_nature-like filter
    _check(oracle, ix, corpus, q, 50, allow, expect_path=1)
    few = np.zeros(16_919, dtype=bool); few[[5, 16_918, 4000]] = True                          # fewer rows pass than n_results
    _check(oracle, ix, corpus, q, 50, few, expect_path=1)
    ix.close()


@pytest.mark.parametrize("d,n,b,k", [(1024, 100_000, 64, 10), (1024, 70_000, 1, 50), (64, 9000, 40, 10), (200, 30_011, 33, 20),
                                     (768, 50_000, 64, 100), (2048, 20_000, 7, 10), (128, 257, 3, 5)])
def test_split_k_bootstrap(eng, oracle, d, n, b, k):
    """option split_boot (default 1): <= 64 queries with a small threshold sample take it with k_boot — one 32-row block per
    workgroup, k-steps dealt to the waves, partial sums added in LDS. Ids and score bits are the oracle's with it and without it,
    with and without a row bitmap (its per-block word, the ragged last block); the sample is the same size or larger."""
    corpus = synth.make_corpus(n, d)
    q = synth.make_queries(b, d, corpus)
    allow = np.random.default_rng(n).random(n) < 0.3
    seen = {}
    for sb in (1, 0):
        ix = _index(eng, corpus, force_fast=1, split_boot=sb)
        st = _check(oracle, ix, corpus, q, k, expect_path=0)
        assert st["exact_queries"] == 0 and st["retried_queries"] == 0, (sb, st)
        _check(oracle, ix, corpus, q, k, allow, expect_path=0)
        ix.set_option("small_scan", 0)                   # the streaming tile kernel as the main scan behind either bootstrap
        st2 = _check(oracle, ix, corpus, q, k, expect_path=0)
        _check(oracle, ix, corpus, q, k, allow, expect_path=0)
        assert st2["exact_queries"] == 0 and st2["retried_queries"] == 0, (sb, st2)
        ix.set_option("cand_cap", 8)                     # segments of 8 slots: overflow -> fallback passes, on both main scans
        for ss in (1, 0):
            ix.set_option("small_scan", ss)
            _check(oracle, ix, corpus, q, k, expect_path=0)
        seen[sb] = st
        ix.close()
    assert seen[1]["sample_rows"] <= max(4 * 256 * 32, seen[0]["sample_rows"]), seen
    assert seen[1]["sample_rows"] % 32 == 0 and seen[1]["sample_rows"] >= min(n // 32 * 32, 8192) or n < 8192, seen


def test_small_scan_path_through_ties(eng, oracle):
    """the split-K path of small searches (k_boot, k_scan_small, k_refine on short candidate lists) when the k-th score is shared by
    identical rows: 8 copies of a row near every query, k = 5, 8 and 12 (ties go to the lowest row id), and k above the number of
    rows a bitmap leaves. Same ids and score bits as the oracle, with the split-K main scan and with the tile kernel."""
    n, d, b = 50_000, 256, 16
    rng = np.random.default_rng(31)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    for j in range(b):
        twin = (q[j] + 0.05 * rng.standard_normal(d)).astype(np.float32)
        corpus[rng.choice(n, size=8, replace=False)] = twin
    allow = np.zeros(n, dtype=bool)
    allow[rng.choice(n, size=40, replace=False)] = True
    for ss in (1, 0):
        ix = _index(eng, corpus, force_fast=1, small_scan=ss)
        for k in (5, 8, 12):
            st = _check(oracle, ix, corpus, q, k, expect_path=0)
            assert st["exact_queries"] == 0, (ss, k, st)
        _check(oracle, ix, corpus, q, 50, allow, expect_path=0)       # 40 allowed rows, k = 50
        ix.close()


def test_speculative_threshold_is_verified(eng, oracle):
    """option spec_tau (default on): the scan threshold comes from a rank below k of the sampled scores — an estimate, not a bound —
    and k_refine verifies it per query. (1) On a random corpus the rank IS below k and the answers are the oracle's. (2) A corpus
    built to fool the estimate: the only rows near the query all sit in tile 0, which every sample contains — the sample's 7th
    best is then far above the corpus' k-th best; verification fails, the queries take the fallback passes (proven threshold),
    the answers are still the oracle's, and the index stops speculating for a while."""
    n, d, k = 400_000, 128, 10
    rng = np.random.default_rng(77)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((70, d)).astype(np.float32)
    ix = _index(eng, corpus)
    st = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert 1 <= st["tau_rank"] < k and st["retried_queries"] == 0 and st["exact_queries"] == 0, st
    ix.set_option("spec_tau", 0)
    st0 = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert st0["tau_rank"] == k and st0["emitted"] > 1.2 * st["emitted"], (st0, st)      # what the speculation saves
    ix.set_option("spec_tau", 1)
    ix.close()
    fool = corpus.copy()
    for j in range(8):                                     # 8 near-copies of each of the first 3 queries in tile 0, one per wave's 32-row
        fool[32 * j: 32 * j + 3] = q[:3] + 0.05 * rng.standard_normal((3, d)).astype(np.float32)   # block (8 distinct bootstrap sets); nothing else is close
    ix = _index(eng, fool, spread_boot=0)                  # (the whole-tile sample this corpus is built against: tile 0 = blocks 0..7)
    st = _check(oracle, ix, fool, q, k, expect_path=0)
    assert st["tau_rank"] < k and st["retried_queries"] >= 3, st
    st = _check(oracle, ix, fool, q, k, expect_path=0)     # backoff: proven thresholds now, nobody retried
    assert st["tau_rank"] == k and st["retried_queries"] == 0, st
    ix.close()


@pytest.mark.parametrize("k", [256, 300])
def test_large_k(eng, oracle, k):
    """k = 256 is the largest the MFMA path serves, k = 300 goes through the exact full scan."""
    corpus = synth.make_corpus(12000, 1024)
    q = synth.make_queries(20, 1024, corpus)
    ix = _index(eng, corpus, force_fast=1)
    st = _check(oracle, ix, corpus, q, k)
    assert st["path"] == (0 if k <= 256 else 1)


def test_single_query_large_corpus_uses_scan(eng, oracle):
    """the reference's production shape (B = 1, k = 50) on a corpus big enough for the streaming scan"""
    corpus = synth.make_corpus(70000, 1024)
    q = synth.make_queries(1, 1024, corpus)
    ix = _index(eng, corpus)
    _check(oracle, ix, corpus, q, 50, expect_path=0)


def test_more_queries_than_one_launch(eng, oracle):
    """nq > 4096 is cut into pipeline passes; results per query are independent of the batching"""
    corpus = synth.make_corpus(5000, 256)
    q = np.random.default_rng(4).standard_normal((4100, 256)).astype(np.float32)
    ix = _index(eng, corpus, force_fast=1)
    s, r, c = ix.search(q, 5)
    s1, r1, c1 = ix.search(q[4090:4100], 5)
    np.testing.assert_array_equal(r[4090:4100], r1)
    np.testing.assert_array_equal(s[4090:4100], s1)
    es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q[:64], 5)
    np.testing.assert_array_equal(r[:64], er)
    np.testing.assert_array_equal(s[:64], es)


def test_concurrent_callers(eng, oracle):
    """one index shared by several threads (the Streamlit cache_resource situation): calls serialise internally"""
    import threading
    corpus = synth.make_corpus(8000, 1024)
    ix = _index(eng, corpus, force_fast=1)
    ch = oracle.normalize_rows(corpus)
    errs = []

    def work(seed):
        try:
            q = np.random.default_rng(seed).standard_normal((30 + seed, 1024)).astype(np.float32)
            for _ in range(3):
                s, r, c = ix.search(q, 10)
                es, er, ec = oracle.cosine_topk(ch, q, 10)
                assert (r == er).all() and (s == es).all()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_device_pointer_path_matches_host_path(eng):
    """RDX_DEVICE (torch CUDA tensors, current stream) and RDX_HOST (numpy) give the same bits"""
    import torch
    corpus = synth.make_corpus(9000, 1024)
    q = synth.make_queries(100, 1024, corpus)
    ix = eng.HipIndex(1024)
    ix.add(torch.from_numpy(corpus).cuda())
    ix.set_option("force_fast", 1)
    s, r, c = ix.search(q, 10)
    qd = torch.from_numpy(q).cuda()
    sd = torch.empty((100, 10), dtype=torch.float32, device="cuda")
    rd = torch.empty((100, 10), dtype=torch.int64, device="cuda")
    cd = torch.empty((100,), dtype=torch.int32, device="cuda")
    ix.search_device(qd, 10, sd, rd, cd)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rd.cpu().numpy(), r)
    np.testing.assert_array_equal(sd.cpu().numpy(), s)
    ix.set_option("row_base", 1000)
    s2, r2, c2 = ix.search(q[:3], 10)
    np.testing.assert_array_equal(r2, r[:3] + 1000)


def test_device_queries_are_ordered_on_the_callers_stream(eng):
    """The query tensor is still being produced on the caller's stream (default stream, then a side stream) when the
    search is enqueued: the search must run behind it, not beside it (include/rdx.h: RDX_DEVICE calls run on `stream`)."""
    import torch
    corpus = synth.make_corpus(30000, 256)
    q = synth.make_queries(8, 256, corpus)
    ix = eng.HipIndex(256)
    ix.add(corpus)
    ix.set_option("force_fast", 1)
    s, r, c = ix.search(q, 10)
    base = torch.from_numpy(q).cuda()
    w = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for stream in (torch.cuda.default_stream(), torch.cuda.Stream()):
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            x = w
            for _ in range(40):                      # tens of ms of GPU work in front of the queries
                x = torch.tanh((x @ w) * 1e-2)
            qd = torch.zeros_like(base)
            qd += base + 0.0 * x[0, :1]              # ready only when the chain has finished
            sd = torch.empty((8, 10), dtype=torch.float32, device="cuda")
            rd = torch.empty((8, 10), dtype=torch.int64, device="cuda")
            cd = torch.empty((8,), dtype=torch.int32, device="cuda")
            ix.search_device(qd, 10, sd, rd, cd)
        stream.synchronize()
        np.testing.assert_array_equal(rd.cpu().numpy(), r)
        np.testing.assert_array_equal(sd.cpu().numpy(), s)


def test_sibling_lockstep_variant_is_exact(eng, oracle):
    """option sib_sync selects the scan variant whose query-tile workgroups keep in step (speed/traffic only): same bits"""
    corpus = synth.make_corpus(70000, 1024)
    q = synth.make_queries(700, 1024, corpus)          # 3 query tiles of 256 -> three sibling workgroups per stream
    ix = eng.HipIndex(1024)
    ix.add(corpus)
    es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q, 10)
    for lag in (3, 6):
        ix.set_option("sib_sync", 1)
        ix.set_option("sib_lag", lag)
        s, r, c = ix.search(q, 10)
        assert ix.last_stats()["path"] == 0
        np.testing.assert_array_equal(r, er)
        np.testing.assert_array_equal(s, es)
    ix.set_option("sib_sync", 0)
    s, r, c = ix.search(q, 10)
    np.testing.assert_array_equal(r, er)


def test_zero_and_tiny_query_vectors(eng, oracle):
    """a zero query (norm clamps at 1e-12, every score 0 -> the first k rows by the tie rule), a denormal-scale query
    and a huge-scale query answer exactly as the oracle does, also when the MFMA path is forced (emits everything ->
    overflow -> exact fallback)"""
    corpus = synth.make_corpus(50000, 1024)
    q = synth.make_queries(6, 1024, corpus)
    q[0] = 0.0
    q[1] *= 1e-30
    q[2] *= 1e30
    q[3, 1:] = 0.0            # a single non-zero component
    for force in (0, 1):
        ix = _index(eng, corpus, force_fast=force)
        _check(oracle, ix, corpus, q, 10)
        ix.close()


def test_clustered_rows_get_a_second_mfma_pass(eng, oracle):
    """similar rows stored together (chunks of one document) between two sampled tiles: the sparse threshold sample misses
    them, their candidate segments overflow, and those queries — only those — are answered by a second MFMA pass with
    a denser sample instead of the exact full scan. Results are the oracle's either way."""
    rng = np.random.default_rng(11)
    n, dim, k = 300_000, 256, 10
    corpus = rng.standard_normal((n, dim)).astype(np.float32)
    v = rng.standard_normal(dim).astype(np.float32)
    a, m = 37 * 256, 20 * 256                      # tiles 37..56: none of them is a multiple of the sampling stride (36)
    sigma = rng.uniform(0.2, 0.6, size=(m, 1)).astype(np.float32)
    corpus[a:a + m] = v + sigma * rng.standard_normal((m, dim)).astype(np.float32)
    q = rng.standard_normal((40, dim)).astype(np.float32)
    q[:5] = v + 0.1 * rng.standard_normal((5, dim)).astype(np.float32)
    ix = _index(eng, corpus, force_fast=1, cand_cap=64, split_boot=0, spread_boot=0)   # (whole-tile sample: the stride this corpus is built against)
    st = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert st["retried_queries"] == 5 and st["exact_queries"] == 0, st
    ix.set_option("split_boot", 1)                 # the split-K bootstrap samples 32-row blocks spread over the corpus: four of them
    st = _check(oracle, ix, corpus, q, k, expect_path=0)   # fall into the cluster, the threshold sees it, nothing overflows
    assert st["retried_queries"] == 0 and st["exact_queries"] == 0, st
    ix.set_option("split_boot", 0)
    ix.set_option("spread_boot", 1)                # the tile kernel's sample as every 36th 32-ROW block (round 4): the cluster's 160 blocks hold
    st = _check(oracle, ix, corpus, q, k, expect_path=0)   # four or five of them — the same rows sampled, eight times finer
    assert st["retried_queries"] == 0 and st["exact_queries"] == 0, st
    ix.set_option("spread_boot", 0)
    ix.set_option("retry", 0)                      # without the second chance the same queries pay the exact scan
    st = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert st["retried_queries"] == 0 and st["exact_queries"] == 5, st
    ix.close()


def test_near_duplicate_band_is_rescored_in_place(eng, oracle):
    """a "document" of thousands of chunks closer together than the fp16 coarse pass can tell apart (their scores lie inside the 2E band
    of the k-th): the band holds more rows than k_refine's ranking arrays. Until round 4 such a query paid the exact full scan of the
    whole corpus; now the band's members are re-scored exactly in place and the k best selected (refine_kernel.hpp). Small and large
    batches (both bootstrap forms), exact ties inside the band, a `where` bitmap through the band; the oracle's ids and score bits,
    and no query handed to the exact scan. More than 1024 rows IDENTICAL to the k-th still need the exact scan — and get it."""
    rng = np.random.default_rng(23)
    n, dim, k = 120_000, 256, 10
    corpus = rng.standard_normal((n, dim)).astype(np.float32)
    a, m = 40_000, 3000
    corpus[a:a + m] = corpus[a] + 0.004 * rng.standard_normal((m, dim)).astype(np.float32)
    corpus[a + 11] = corpus[a + 5]
    corpus[a + 2000] = corpus[a + 5]
    for b in (8, 200):
        q = rng.standard_normal((b, dim)).astype(np.float32)
        q[: b // 2] = corpus[a] + 0.05 * rng.standard_normal((b // 2, dim)).astype(np.float32)
        ix = _index(eng, corpus, force_fast=1)
        st = _check(oracle, ix, corpus, q, k, expect_path=0)
        assert st["exact_queries"] == 0, st
        assert st["rescored"] >= (b // 2) * 1025, st                 # the band really was larger than the ranking arrays
        allow = rng.random(n) < 0.5
        st = _check(oracle, ix, corpus, q, k, allow, expect_path=0)
        assert st["exact_queries"] == 0, st
        st = _check(oracle, ix, corpus, q, 100, expect_path=0)
        assert st["exact_queries"] == 0, st
        ix.close()
    corpus[a:a + 1200] = corpus[a]                                   # 1200 identical rows: ties that only the exact scan orders
    q = rng.standard_normal((4, dim)).astype(np.float32)
    q[0] = corpus[a]                                                 # the identical rows ARE the best: the k-th score is shared by 1200 rows
    ix = _index(eng, corpus, force_fast=1)
    st = _check(oracle, ix, corpus, q, k, expect_path=0)
    assert st["exact_queries"] == 1, st
    ix.close()
