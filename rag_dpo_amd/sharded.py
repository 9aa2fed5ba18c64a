"""Row-sharded search over the ranks of one torch.distributed group (one process per GPU).

Rows are independent, so the corpus is cut into contiguous row ranges, one per rank; every rank holds the
same query batch, scans its own shard (librdx) and produces a partial top-k with GLOBAL row ids. The only
exchange step is ONE all-gather of the packed partials (B*k*12 + B*4 bytes per rank: 124 KB at B=1024, k=10
— latency-bound on xGMI, nowhere near a link's bandwidth), after which every rank merges the world*k
candidates per query with the same (score desc, row asc) rule -> bit-identical to a single-GPU search,
because every (query, row) score is computed by the same fixed-order arithmetic wherever the row lives.

Backends: `HipShard` (product: HipIndex + rdx_merge_topk, tensors on cuda, backend nccl = RCCL).
The CPU/gloo tests inject their own backend built on the oracle (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous shard g = rows [g*ceil(N/G), ...) as in SURVEY.md §8e"""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


class HipShard:
    """this rank's shard in HBM + the device-side merge (product backend)"""

    def __init__(self, dim: int, device: int, row_offset: int = 0):
        from .engine import HipIndex
        from . import _lib as L
        self.index = HipIndex(dim, device)
        self.index.set_option("row_base", int(row_offset))   # the shard answers with GLOBAL row ids
        self._L = L
        self._lib = L.load(require_gpu=True)
        self.device = torch.device("cuda", device)
        self.dev_index = device

    def add(self, rows):
        self.index.add(rows)

    def __len__(self):
        return len(self.index)

    def search(self, queries: torch.Tensor, k: int, out_score, out_row, out_count):
        self.index.search_device(queries, k, out_score, out_row, out_count)

    def search_async(self, queries: torch.Tensor, k: int, out_score, out_row, out_count) -> bool:
        """enqueue only (rdx_search_async); False when the batch is too large for the asynchronous form"""
        if queries.shape[0] > 4096:
            return False
        self.index.search_device_async(queries, k, out_score, out_row, out_count)
        return True

    def search_wait(self) -> bool:
        return self.index.search_wait()

    def merge_packed(self, packed: torch.Tensor, part_stride: int, n_parts: int, nq: int, k: int, out_score, out_row, out_count):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        self._L.check(self._lib.rdx_merge_topk_packed(self.dev_index, p(packed), int(part_stride), int(n_parts), int(nq), int(k),
                                                      p(out_score), p(out_row), p(out_count), ctypes.c_void_p(stream)))


class ShardedSearcher:
    """`shard` provides search(queries, k, out_score, out_row, out_count) answering with GLOBAL row ids and
    merge_packed(...) over the all-gather receive buffer."""

    def __init__(self, shard, group: Optional[dist.ProcessGroup] = None, device=None, host_staged: bool = False,
                 always_exchange: bool = False):
        self.shard = shard
        self.group = group
        self.host_staged = host_staged
        self.always_exchange = always_exchange   # world 1: still run the all-gather + merge (exercises RCCL on a one-GPU box)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else getattr(shard, "device", torch.device("cpu"))
        self._bufs = {}
        self._views = {}

    def _buffers(self, nq: int, k: int):
        key = (nq, k)
        if key not in self._bufs:
            per = nq * k * 12 + nq * 4           # rows i64 | scores f32 | counts i32  (include/rdx.h)
            per_pad = (per + 15) // 16 * 16
            local = torch.zeros(per_pad, dtype=torch.uint8, device=self.device)
            allb = torch.zeros(self.world * per_pad, dtype=torch.uint8, device=self.device)
            out = (torch.empty((nq, k), dtype=torch.float32, device=self.device),
                   torch.empty((nq, k), dtype=torch.int64, device=self.device),
                   torch.empty((nq,), dtype=torch.int32, device=self.device))
            self._bufs[key] = (per_pad, local, allb, out)
            self._views[key] = self.views(local, nq, k)   # (ten .view() calls: 12 us of a 60 us search when made per call)
        return self._bufs[key]

    @staticmethod
    def views(buf: torch.Tensor, nq: int, k: int):
        r = buf[: nq * k * 8].view(torch.int64).view(nq, k)
        s = buf[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
        c = buf[nq * k * 12: nq * k * 12 + nq * 4].view(torch.int32)
        return s, r, c

    def search(self, queries: torch.Tensor, k: int, query_src: Optional[int] = None):
        """queries: [nq][dim] fp32 on self.device, identical on every rank; k >= 1. Returns (score, row, count)
        tensors with GLOBAL row ids, identical on every rank.
        query_src = r: only rank r's `queries` holds the batch (the rank that took the request); the other ranks pass a
        tensor of the same shape to receive it — one broadcast (B*d*4 bytes: 4 MB at B = 1024) in front of the scan
        (SURVEY.md §8e)."""
        self.search_begin(queries, k, query_src)
        return self.search_end()

    def search_begin(self, queries: torch.Tensor, k: int, query_src: Optional[int] = None) -> None:
        """everything of search() that is ENQUEUED: the shard's search, the exchange step and the merge go onto the current
        stream and the call returns; search_end() waits and hands out the result. Work the caller enqueues on the same stream in
        between (the encode of its next query batch, BASELINE config 5) runs right behind the search without waiting for the host."""
        if k < 1:
            raise ValueError("k must be >= 1")
        if getattr(self, "_open", None) is not None:
            raise RuntimeError("search_begin() twice without search_end()")
        nq = queries.shape[0]
        if query_src is not None and self.world > 1:
            if self.host_staged:
                h_q = queries.cpu()
                dist.broadcast(h_q, src=query_src, group=self.group)
                queries.copy_(h_q)
            else:
                dist.broadcast(queries, src=query_src, group=self.group)
        per_pad, local, allb, out = self._buffers(nq, k)
        s, r, c = self._views[(nq, k)]
        # The shard's search is ENQUEUED, the exchange step is enqueued right behind it on the same stream, and only then does
        # the host wait for the search (rdx_search_wait): the host-side cost of launching the collective overlaps the scan
        # instead of leaving the GPU idle after it. If the search had to re-run overflowed queries (rare) the exchange is repeated.
        deferred = hasattr(self.shard, "search_async") and self.shard.search_async(queries, k, s, r, c)
        if not deferred:
            self.shard.search(queries, k, s, r, c)
        exchange = not (self.world == 1 and not (self.always_exchange and dist.is_initialized()))
        if exchange:
            self._exchange(nq, k)
        self._open = (nq, k, deferred, exchange)

    def _exchange(self, nq: int, k: int) -> None:
        per_pad, local, allb, out = self._buffers(nq, k)
        if self.host_staged:
            # rehearsal only (several ranks sharing ONE GPU over gloo, which cannot move device memory): same packed
            # layout, same merge kernel, the collective alone goes through host memory
            h_all = torch.empty(self.world * per_pad, dtype=torch.uint8)
            dist.all_gather_into_tensor(h_all, local.cpu(), group=self.group)
            allb.copy_(h_all)
        else:
            dist.all_gather_into_tensor(allb, local, group=self.group)   # the ONE exchange step (RCCL over xGMI)
        self.shard.merge_packed(allb, per_pad, self.world, nq, k, out[0], out[1], out[2])

    def search_end(self):
        if getattr(self, "_open", None) is None:
            raise RuntimeError("search_end() without search_begin()")
        nq, k, deferred, exchange = self._open
        self._open = None
        per_pad, local, allb, out = self._buffers(nq, k)
        if not exchange:
            if deferred:
                self.shard.search_wait()
            return self._views[(nq, k)]
        if deferred and self._any_redone(self.shard.search_wait()):
            self._exchange(nq, k)   # some rank's fallback passes rewrote its partial after the first exchange: exchange and merge again
        return out

    def _any_redone(self, mine: bool) -> bool:
        """every rank must take the same branch: a one-int all-reduce, only when the asynchronous form is in use. The common
        answer (nobody) costs one tiny collective per search; ranks whose searches never overflow still pay it."""
        if self.world == 1:
            return mine
        flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device="cpu" if self.host_staged else self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return bool(flag.item())
