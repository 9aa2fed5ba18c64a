"""BASELINE.json's full sizes (configs 3 and 4: 1M x 1024 / B=256 / k=100 and 10M x 1024 / B=1024 / k=10 on one GPU),
checked through properties that do not need a CPU pass over the whole corpus:

  order        every result list is sorted by (score desc, row asc) and holds k distinct rows
  exact score  every returned (row, score) equals the oracle's score of the STORED row, bit for bit
  known answer a query that IS a stored row comes back first with the score of the row against itself; a query planted
               next to a row finds that row first
  partition    searching the two halves of the corpus separately (row bitmaps) and merging the partial lists with
               rdx_merge_topk gives the full search again, bit for bit (what the multi-GPU path relies on)
  sample       restricted by the row bitmap to 128K rows, ids and scores equal the oracle's exact top-k of those rows
  full scan    the exact full scan K5 (fp32 master, fp64 lane-order sums, no MFMA, no threshold heuristics; itself checked
               against the oracle at small N in test_gpu_parity.py) over ALL rows returns the same ids and score bits as
               the MFMA path for 32 queries: no better row exists anywhere in the corpus

The corpus and the queries follow SURVEY.md §8(d): 1 % exact duplicate rows, 10 % of the queries planted next to a row.

The oracle is only the checker here. ~41 GB + 20 GB of HBM at 10M rows."""
import numpy as np
import pytest

from rag_dpo_amd import synth

pytestmark = pytest.mark.gpu


def _build(eng, rows, dim):
    import torch
    ix = eng.HipIndex(dim)
    ix.reserve(rows)
    for j, r0 in enumerate(range(0, rows, synth.CHUNK)):
        ix.add(synth.torch_corpus_chunk(j, min(synth.CHUNK, rows - r0), dim, "cuda:0"))
    torch.cuda.synchronize()
    return ix


@pytest.mark.parametrize("rows,b,k", [(1_000_000, 256, 100), (10_000_000, 1024, 10)])
def test_full_size_properties(rows, b, k, oracle):
    import torch
    from rag_dpo_amd import engine as eng
    dim = 1024
    ix = _build(eng, rows, dim)
    assert len(ix) == rows
    rng = np.random.default_rng(7)
    q, planted = synth.torch_queries(b, dim, "cuda:0", total_rows=rows, return_planted=True)
    q = q.cpu().numpy()
    # known answers: query 0..7 ARE stored rows (taken back out of the index), query 8..15 are planted next to rows
    known = rng.integers(0, rows, size=16)
    stored = ix.get(known)
    q[:8] = stored[:8]
    q[8:16] = stored[8:16] + (0.2 / np.sqrt(dim)) * rng.standard_normal((8, dim)).astype(np.float32)
    s, r, c = ix.search(q, k)
    st = ix.last_stats()
    assert st["path"] == 0 and st["exact_queries"] == 0, st          # the MFMA scan answered, no fallback
    assert (c == k).all()
    # order + distinct
    sd = s.astype(np.float64)
    assert (np.diff(sd, axis=1) <= 0).all()
    ties = np.diff(sd, axis=1) == 0
    assert (np.diff(r, axis=1)[ties] > 0).all()
    assert all(len(set(row.tolist())) == k for row in r)
    # exact score of every returned row (checked on 64 queries: 64*k rows fetched back)
    qhat = oracle.normalize_rows(q)
    for i in list(range(16)) + rng.choice(np.arange(16, b), size=48, replace=False).tolist():
        np.testing.assert_array_equal(oracle.scores(ix.get(r[i]), qhat[i]), s[i])
    # known answers
    for i in range(16):
        first = ix.get(r[i, :1])[0]
        assert (first == stored[i]).all(), i          # the row itself, or an identical row with a lower id
        assert r[i, 0] <= known[i]
    assert (s[:8, 0] > 0.999999).all()
    # §8(d) planted queries (q = c_i + 0.3 eps): the row comes back first, or an exact duplicate of it with a lower id
    for qi, ri in planted:
        if qi >= 16:
            if r[qi, 0] != ri:
                a, b_ = ix.get(np.array([r[qi, 0], ri]))
                assert r[qi, 0] < ri and (a == b_).all(), (qi, ri, r[qi, 0])
    # full scan: K5 over all rows == the MFMA path, for the known-answer queries, planted ones and plain random ones
    nx = 32
    pick = np.array(list(range(16)) + [qi for qi, _ in planted if qi >= 16][:8], dtype=np.int64)
    pick = np.concatenate([pick, np.setdiff1d(np.arange(16, b), pick)[: nx - len(pick)]])
    ix.set_option("force_exact", 1)
    xs, xr, xc = ix.search(q[pick], k)
    assert ix.last_stats()["path"] == 1
    ix.set_option("force_exact", 0)
    np.testing.assert_array_equal(xr, r[pick])
    np.testing.assert_array_equal(xs, s[pick])
    np.testing.assert_array_equal(xc, c[pick])
    # partition: two halves by bitmap, merged = full
    nchk = 32
    half = np.zeros(rows, dtype=bool)
    half[: rows // 2 + 12345] = True
    pa = ix.search(q[:nchk], k, oracle.pack_mask(half, rows))
    pb = ix.search(q[:nchk], k, oracle.pack_mask(~half, rows))
    ms, mr, mc = eng.merge_topk(np.stack([pa[0], pb[0]]), np.stack([pa[1], pb[1]]), np.stack([pa[2], pb[2]]), k)
    np.testing.assert_array_equal(mr, r[:nchk])
    np.testing.assert_array_equal(ms, s[:nchk])
    # sample: exact oracle top-k of 128K rows scattered over the corpus
    rows_s = np.sort(rng.choice(rows, size=131072, replace=False))
    allow = np.zeros(rows, dtype=bool)
    allow[rows_s] = True
    gs, gr, gc = ix.search(q[:16], k, oracle.pack_mask(allow, rows))
    es, er, ec = oracle.cosine_topk(ix.get(rows_s), q[:16], k)
    np.testing.assert_array_equal(gr, rows_s[er])
    np.testing.assert_array_equal(gs, es)
    ix.close()
    torch.cuda.empty_cache()


def test_config5_as_a_chain_full_size(oracle):
    """BASELINE config 5 end to end on one GPU: 1024 query TEXTS -> `embed_device` (XLM-R-large architecture, random-init fp16: the
    packed forward with librdx's kernels) -> straight into the search of a 10 M-row bf16 corpus held at 4 B/element (`compact_master`:
    raw bf16 rows + one divisor per row), nothing leaving the device in between. Encoder VALUES are unpinned (no BGE-M3 weights
    offline); what is checked is the chain: whatever vectors the encoder hands over, the search must return their exact neighbours.
    Same properties as above — order, the oracle's score of every returned stored row, the exact full scan K5 over ALL rows for 32
    queries, the exact oracle top-k of a 128 K-row sample — plus: handing the same vectors over as host floats gives the same bits."""
    import torch
    from rag_dpo_amd import engine as eng
    from rag_dpo_amd.embedding_provider import EmbeddingProvider
    rows, b, k, dim = 10_000_000, 1024, 10, 1024
    ix = eng.HipIndex(dim)
    ix.set_option("compact_master", 1)
    ix.reserve(rows)
    for j, r0 in enumerate(range(0, rows, synth.CHUNK)):
        ix.add_bf16(synth.torch_corpus_chunk(j, min(synth.CHUNK, rows - r0), dim, "cuda:0").to(torch.bfloat16))
    torch.cuda.synchronize()
    assert len(ix) == rows
    p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=1024).load()
    texts = synth.query_texts(b)
    qd = p.embed_device(texts)                                       # [1024][1024] fp32 on the device, not normalised (the search runs K1)
    assert qd.is_cuda and qd.shape == (b, dim) and bool(torch.isfinite(qd).all())
    s_d = torch.empty((b, k), dtype=torch.float32, device="cuda")
    r_d = torch.empty((b, k), dtype=torch.int64, device="cuda")
    c_d = torch.empty((b,), dtype=torch.int32, device="cuda")
    ix.search_device(qd, k, s_d, r_d, c_d)
    torch.cuda.synchronize()
    st = ix.last_stats()
    assert st["path"] == 0, st
    s, r, c, q = s_d.cpu().numpy(), r_d.cpu().numpy(), c_d.cpu().numpy(), qd.cpu().numpy()
    assert (c == k).all()
    sd = s.astype(np.float64)
    assert (np.diff(sd, axis=1) <= 0).all()
    assert (np.diff(r, axis=1)[np.diff(sd, axis=1) == 0] > 0).all()
    assert all(len(set(row.tolist())) == k for row in r)
    rng = np.random.default_rng(5)
    qhat = oracle.normalize_rows(q)
    for i in rng.choice(b, size=64, replace=False).tolist():         # the oracle's score of every returned STORED row (bf16 values, fp64 divisor)
        np.testing.assert_array_equal(oracle.scores(ix.get(r[i]), qhat[i]), s[i])
    pick = np.sort(rng.choice(b, size=32, replace=False))
    ix.set_option("force_exact", 1)                                  # K5: fp32 arithmetic on the recomputed normalised rows, no MFMA, all rows
    xs, xr, xc = ix.search(q[pick], k)
    assert ix.last_stats()["path"] == 1
    ix.set_option("force_exact", 0)
    np.testing.assert_array_equal(xr, r[pick])
    np.testing.assert_array_equal(xs, s[pick])
    rows_s = np.sort(rng.choice(rows, size=131072, replace=False))
    allow = np.zeros(rows, dtype=bool)
    allow[rows_s] = True
    gs, gr, gc = ix.search(q[:16], k, oracle.pack_mask(allow, rows))
    es, er, ec = oracle.cosine_topk(ix.get(rows_s), q[:16], k)
    np.testing.assert_array_equal(gr, rows_s[er])
    np.testing.assert_array_equal(gs, es)
    # the reference's hand-over (the vectors as host floats, src/utils/embedding_provider.py:147) answers exactly like the device hand-over
    hs, hr, hc = ix.search(q[:8], k)
    np.testing.assert_array_equal(hr, r[:8])
    np.testing.assert_array_equal(hs, s[:8])
    p.unload()
    ix.close()
    torch.cuda.empty_cache()
