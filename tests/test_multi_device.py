"""rag_dpo_amd.multi_device: the rows of ONE collection over several devices in ONE process. CPU: the orchestration
(water-filled placement, row-id maps, per-shard bitmaps, compaction, rollback, one merge whatever D x k) with TEST-ONLY oracle
shards standing in for the per-device HipIndex — every result must equal a single oracle engine holding all rows.
GPU (one card): three HipIndex shards on cuda:0 must equal one HipIndex bit for bit."""
import numpy as np
import pytest

from rag_dpo_amd import synth
from rag_dpo_amd.collection import Collection
from rag_dpo_amd.multi_device import MultiDeviceIndex

from oracle_engine import OracleEngine


class OracleShard(OracleEngine):
    """an oracle engine that answers with mapped row ids, as HipIndex does after rdx_index_set_row_ids"""

    def __init__(self, dim, device=0):
        super().__init__(dim, device)
        self.ids = np.zeros(0, dtype=np.int64)

    def set_row_ids(self, first, ids):
        ids = np.asarray(ids, dtype=np.int64)
        assert first + ids.shape[0] <= len(self) and (np.diff(ids) > 0).all()
        if self.ids.shape[0] < len(self):
            self.ids = np.concatenate([self.ids, np.arange(self.ids.shape[0], len(self))])
        self.ids[first:first + ids.shape[0]] = ids

    def compact(self, keep):
        super().compact(keep)
        self.ids = np.zeros(0, dtype=np.int64)       # like the library: compaction drops the map

    def search(self, q, k, allow_bits=None):
        s, r, c = super().search(q, k, allow_bits)
        ids = self.ids if self.ids.shape[0] == len(self) else np.arange(len(self))
        return s, np.where(r >= 0, ids[np.clip(r, 0, None)] if len(self) else r, -1), c


def oracle_merge(ps, pr, pc, k, device):
    from oracle import oracle as O
    return O.merge_topk(ps, pr, pc, k)


def make(dim, n_dev):
    return MultiDeviceIndex(dim, list(range(n_dev)), shard_factory=OracleShard, merge=oracle_merge)


def same(a, b):
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_multi_device_equals_single_engine(n_dev, oracle):
    rng = np.random.default_rng(n_dev)
    dim = 64
    md, one = make(dim, n_dev), OracleEngine(dim)
    corpus = synth.make_corpus(3000, dim)          # 1 % duplicate rows: ties across shards must order by collection row id
    a = 0
    for step in [1, 100, 100, 7, 1000, 64, 100, 1628]:     # the indexer's batches of 100, odd sizes, one big one
        md.add(corpus[a:a + step]); one.add(corpus[a:a + step]); a += step
    assert len(md) == len(one) == 3000
    sizes = [len(s) for s in md._shards]
    assert max(sizes) - min(sizes) <= 1, sizes                # water-filling keeps the shards level
    q = synth.make_queries(9, dim, corpus)
    for k in (1, 10, 50, 700, 2500):                          # 700 * 8, 2500 * 2 > 4096: more candidates than one merge launch ranks
        same(md.search(q, k), one.search(q, k))
    same([md.get(np.array([5, 2999, 0, 1234]))], [one.get(np.array([5, 2999, 0, 1234]))])
    # where bitmaps: collection bitmap -> per-shard bitmaps; resident form too
    allow = rng.random(3000) < 0.3
    bits = oracle.pack_mask(allow, 3000)
    same(md.search(q, 20, bits), one.search(q, 20, bits))
    m = md.make_mask(bits)
    same(md.search(q, 20, mask=m), one.search(q, 20, bits))
    same(md.search(q, 20, oracle.pack_mask(np.zeros(3000, bool), 3000)), one.search(q, 20, oracle.pack_mask(np.zeros(3000, bool), 3000)))
    # update in place
    ids = np.array([3, 1500, 2998, 77])
    new = rng.standard_normal((4, dim)).astype(np.float32)
    md.update(ids, new); one.update(ids, new)
    same(md.search(new, 5), one.search(new, 5))
    # compaction renumbers rows 0..n_keep-1 on both sides
    keep = np.flatnonzero(rng.random(3000) < 0.6)
    md.compact(keep); one.compact(keep)
    assert len(md) == len(one) == keep.shape[0]
    same(md.search(q, 30), one.search(q, 30))
    with pytest.raises(ValueError):
        md.search(q, 5, mask=m)                               # a mask does not outlive a write
    # adds after a compaction go to the emptiest shards and keep global ids appended
    extra = rng.standard_normal((500, dim)).astype(np.float32)
    md.add(extra); one.add(extra)
    same(md.search(q, 30), one.search(q, 30))
    same(md.search(extra[:4], 3), one.search(extra[:4], 3))
    # a rejected batch stores nothing on any shard
    bad = rng.standard_normal((300, dim)).astype(np.float32)
    bad[250, 3] = np.nan
    n0, sizes0 = len(md), [len(s) for s in md._shards]
    with pytest.raises(ValueError):
        md.add(bad)
    assert len(md) == n0 and [len(s) for s in md._shards] == sizes0
    same(md.search(q, 30), one.search(q, 30))
    md.close()


def test_collection_over_several_devices_cpu():
    """the whole Chroma-shaped contract (ids, where, paging, tombstones, compaction, mask cache) on a 3-"device" engine"""
    from test_collection import run_collection_contract, run_mask_cache
    factory = lambda dim, device=0: make(dim, 3)
    col = run_collection_contract(factory)
    assert isinstance(col._engine, MultiDeviceIndex)
    run_mask_cache(factory)


def test_empty_and_tiny():
    md = make(64, 4)
    q = np.ones((2, 64), np.float32)
    s, r, c = md.search(q, 5)
    assert (c == 0).all() and (r == -1).all()
    md.add(np.eye(64, dtype=np.float32)[:1])                 # one row: three shards stay empty
    s, r, c = md.search(q, 5)
    assert (c == 1).all() and (r[:, 0] == 0).all()


@pytest.mark.gpu
def test_three_shards_on_one_gpu_equal_one_index(oracle):
    """three HipIndex shards (all on cuda:0: this box has one card) behind MultiDeviceIndex == one HipIndex, ids and score
    bits, through row-id maps, per-shard resident masks, compaction and the device merge kernel"""
    from rag_dpo_amd.engine import HipIndex
    rng = np.random.default_rng(5)
    dim, n = 256, 30_000
    corpus = synth.make_corpus(n, dim)
    q = synth.make_queries(70, dim, corpus)
    md, one = MultiDeviceIndex(dim, [0, 0, 0]), HipIndex(dim)
    for a in range(0, n, 7000):
        md.add(corpus[a:a + 7000]); one.add(corpus[a:a + 7000])
    for ix in (one, md):
        ix.set_option("force_fast", 1)
    same(md.search(q, 10), one.search(q, 10))
    es, er, ec = oracle.cosine_topk(oracle.normalize_rows(corpus), q, 10)
    same(md.search(q, 10), (es, er, ec))
    allow = rng.random(n) < 0.2
    bits = oracle.pack_mask(allow, n)
    m = md.make_mask(bits)
    same(md.search(q, 50, mask=m), one.search(q, 50, bits))
    for ix in (one, md):
        ix.set_option("force_fast", 0); ix.set_option("force_exact", 1)
    same(md.search(q[:5], 300), one.search(q[:5], 300))       # exact path through the id map as well
    same(md.search(q[:5], 2100), one.search(q[:5], 2100))     # 3 x 2100 candidates per query: the library folds the parts pairwise
    # device in / device out (queries on the first device, peer copies, asynchronous shard searches, ONE device merge)
    import torch
    for ix in (one, md):
        ix.set_option("force_exact", 0); ix.set_option("force_fast", 1)
    qd = torch.from_numpy(q).cuda()
    for kk in (10, 300, 2100):                                # 3 x 2100 candidates per query: folded pairwise on the device
        hs, hr, hc = md.search(q[:9] if kk > 300 else q, kk)
        ds, dr, dc = md.search_device(qd[:9].contiguous() if kk > 300 else qd, kk)
        torch.cuda.synchronize()
        same((ds.cpu().numpy(), dr.cpu().numpy(), dc.cpu().numpy()), (hs, hr, hc))
    ds, dr, dc = md.search_device(qd, 50, mask=m)
    torch.cuda.synchronize()
    same((ds.cpu().numpy(), dr.cpu().numpy(), dc.cpu().numpy()), one.search(q, 50, bits))
    for ix in (one, md):
        ix.set_option("force_fast", 0); ix.set_option("force_exact", 1)
    keep = np.flatnonzero(rng.random(n) < 0.5)
    md.compact(keep); one.compact(keep)
    for ix in (one, md):
        ix.set_option("force_exact", 0); ix.set_option("force_fast", 1)
    same(md.search(q, 10), one.search(q, 10))
    same([md.get(np.array([0, 17, len(md) - 1]))], [one.get(np.array([0, 17, len(one) - 1]))])
    # behind the boundary
    col = Collection("c", devices=[0, 0])
    col.add(ids=[f"r{i}" for i in range(2000)], embeddings=corpus[:2000], metadatas=[{"p": i % 3} for i in range(2000)])
    assert isinstance(col._engine, MultiDeviceIndex)
    ref = Collection("c1")
    ref.add(ids=[f"r{i}" for i in range(2000)], embeddings=corpus[:2000], metadatas=[{"p": i % 3} for i in range(2000)])
    qa = q[:3].tolist()
    assert col.query(query_embeddings=qa, n_results=20, where={"p": 1}) == ref.query(query_embeddings=qa, n_results=20, where={"p": 1})
    # the same through the device entry of the boundary: distances, rows -> ids
    for c_ in (col, ref):
        dist, rows, cnt = c_.query_device(torch.from_numpy(q[:3]).cuda(), n_results=20, where={"p": 1})
        torch.cuda.synchronize()
        want = ref.query(query_embeddings=qa, n_results=20, where={"p": 1})
        assert c_.ids_of(rows.cpu()) == want["ids"] and cnt.tolist() == [20, 20, 20]
        assert [[float(x) for x in row] for row in dist.cpu().numpy()] == want["distances"]
    md.close(); one.close()
