"""Row-sharded search over the ranks of one torch.distributed group (one process per GPU).

Rows are independent, so the corpus is cut into contiguous row ranges, one per rank; every rank holds the
same query batch, scans its own shard (librdx) and produces a partial top-k with GLOBAL row ids. The only
exchange step is ONE all-gather of the packed partials (B*k*12 + B*4 + 16 bytes per rank: 124 KB at B=1024, k=10
— latency-bound on xGMI, nowhere near a link's bandwidth), after which every rank merges the world*k
candidates per query with the same (score desc, row asc) rule -> bit-identical to a single-GPU search,
because every (query, row) score is computed by the same fixed-order arithmetic wherever the row lives.

No second collective and no device-to-host copy per step: a shard whose first pass left some queries incomplete
(candidate overflow, rare) says so in the flags word of its packed partial; the word travels with the all-gather,
the merge kernel ORs the ranks' words into a pinned signal, and every rank reads the SAME answer from it — only
then is the exchange repeated (include/rdx.h: rdx_search_async(out_flags), rdx_signal).

Backends: `HipShard` (product: HipIndex + rdx_merge_topk_packed, tensors on cuda, backend nccl = RCCL).
The CPU/gloo tests inject their own backend built on the oracle (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
import torch.distributed as dist

PACKED_FLAGS = 4   # include/rdx.h RDX_PACKED_FLAGS


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous shard g = rows [g*ceil(N/G), ...) as in SURVEY.md §8e"""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def packed_bytes(nq: int, k: int) -> int:
    """one rank's contribution to the all-gather: rows i64 | scores f32 | counts i32 | flags i32[4] (include/rdx.h)"""
    return nq * k * 12 + nq * 4 + 4 * PACKED_FLAGS


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
        self._sig = ctypes.c_void_p()
        L.check(self._lib.rdx_signal_create(int(device), ctypes.byref(self._sig)))
        self._sig_stream = None

    def close(self):
        if getattr(self, "_sig", None) is not None and self._sig.value:
            self._lib.rdx_signal_destroy(self._sig)
            self._sig = ctypes.c_void_p()
        self.index.close()

    def __del__(self):
        try:
            if getattr(self, "_sig", None) is not None and self._sig.value:
                self._lib.rdx_signal_destroy(self._sig)
                self._sig = ctypes.c_void_p()
        except Exception:
            pass

    def add(self, rows):
        self.index.add(rows)

    def __len__(self):
        return len(self.index)

    def search(self, queries: torch.Tensor, k: int, out_score, out_row, out_count):
        self.index.search_device(queries, k, out_score, out_row, out_count)

    def search_async(self, queries: torch.Tensor, k: int, out_score, out_row, out_count, out_flags) -> bool:
        """enqueue only (rdx_search_async); False when the batch is too large for the asynchronous form"""
        if queries.shape[0] > 4096:
            return False
        self.index.search_device_async(queries, k, out_score, out_row, out_count, out_flags)
        return True

    def search_wait(self) -> bool:
        return self.index.search_wait()

    def merge_packed(self, packed: torch.Tensor, part_stride: int, n_parts: int, nq: int, k: int, out_score, out_row, out_count):
        """enqueue the merge of the gathered partials; its first block publishes the OR of their flags words (merge_flag)"""
        stream = self.index._raw_stream(self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        self._L.check(self._lib.rdx_merge_topk_packed(self.dev_index, p(packed), int(part_stride), int(n_parts), int(nq), int(k),
                                                      p(out_score), p(out_row), p(out_count), self._sig, ctypes.c_void_p(stream)))
        self._sig_stream = stream

    def merge_flag(self) -> bool:
        """True when some rank's partial of the LAST merge_packed carried the "incomplete" flag (same answer on every rank)"""
        v = ctypes.c_int32(0)
        self._L.check(self._lib.rdx_signal_wait(self._sig, ctypes.c_void_p(self._sig_stream), ctypes.byref(v)))
        return bool(v.value)


class ShardedSearcher:
    """`shard` provides search(queries, k, out_score, out_row, out_count) answering with GLOBAL row ids,
    merge_packed(...) over the all-gather receive buffer and merge_flag(); optionally the asynchronous pair
    search_async(..., out_flags) / search_wait()."""

    MERGE_MAX = 4096   # candidates per query rdx_merge_topk_packed ranks in one block (include/rdx.h): world * k must not exceed it

    def __init__(self, shard, group: Optional[dist.ProcessGroup] = None, device=None, host_staged: bool = False,
                 always_exchange: bool = False):
        self.shard = shard
        self.group = group
        self.host_staged = host_staged
        self.always_exchange = always_exchange   # world 1: still run the all-gather + merge (a one-rank RCCL group on a one-GPU box;
                                                 # without a process group the gather of the one part is a copy)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else getattr(shard, "device", torch.device("cpu"))
        self._bufs = {}
        self._views = {}
        self._open = None
        self.exchanges = 0            # exchange steps run so far (a repeated one counts)
        self.time_events = False      # True: record events around the pieces of a step (breakdown(); costs a few us per step)
        self._ev = []

    def _buffers(self, nq: int, k: int):
        key = (nq, k)
        if key not in self._bufs:
            per = packed_bytes(nq, k)
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
        """(scores, rows, counts, flags) views of one packed partial"""
        r = buf[: nq * k * 8].view(torch.int64).view(nq, k)
        s = buf[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
        c = buf[nq * k * 12: nq * k * 12 + nq * 4].view(torch.int32)
        f = buf[nq * k * 12 + nq * 4: nq * k * 12 + nq * 4 + 4 * PACKED_FLAGS].view(torch.int32)
        return s, r, c, f

    def broadcast_queries(self, queries: torch.Tensor, src: int = 0) -> torch.Tensor:
        """the rank that took the request hands the batch to the others: one broadcast of B*d*4 bytes (4 MB at B = 1024,
        SURVEY.md §8e). Every rank then searches the SAME bits (nothing depends on each rank re-deriving the batch)."""
        if self.world > 1:
            if self.host_staged:
                h_q = queries.cpu()
                dist.broadcast(h_q, src=src, group=self.group)
                queries.copy_(h_q)
            else:
                dist.broadcast(queries, src=src, group=self.group)
        return queries

    def search(self, queries: torch.Tensor, k: int, query_src: Optional[int] = None):
        """queries: [nq][dim] fp32 on self.device, identical on every rank; k >= 1. Returns (score, row, count)
        tensors with GLOBAL row ids, identical on every rank.
        query_src = r: only rank r's `queries` holds the batch (the rank that took the request); the other ranks pass a
        tensor of the same shape to receive it — one broadcast in front of the scan."""
        self.search_begin(queries, k, query_src)
        return self.search_end()

    def _mark(self):
        if self.time_events and self.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev[-1].append(e)

    def search_begin(self, queries: torch.Tensor, k: int, query_src: Optional[int] = None) -> None:
        """everything of search() that is ENQUEUED: the shard's search, the exchange step and the merge go onto the current
        stream and the call returns; search_end() waits and hands out the result. Work the caller enqueues on the same stream in
        between (the encode of its next query batch, BASELINE config 5) runs right behind the search without waiting for the host.
        The searcher keeps `queries` alive until search_end(); the caller may overwrite the tensor with stream-ordered work."""
        if k < 1:
            raise ValueError("k must be >= 1")
        if self._open is not None:
            raise RuntimeError("search_begin() twice without search_end()")
        if (self.world > 1 or self.always_exchange) and self.world * k > self.MERGE_MAX:
            # checked BEFORE anything is enqueued: a rank that failed in the merge, behind an all-gather every other rank has
            # entered, would leave them waiting (rdx_merge_topk_packed ranks world * k candidates per query in one block)
            raise ValueError(f"sharded search: world * k = {self.world * k} exceeds {self.MERGE_MAX} candidates per query "
                             f"(k <= {self.MERGE_MAX // self.world} at world {self.world}); use MultiDeviceIndex, whose merge folds")
        nq = queries.shape[0]
        if query_src is not None:
            self.broadcast_queries(queries, query_src)
        per_pad, local, allb, out = self._buffers(nq, k)
        s, r, c, f = self._views[(nq, k)]
        if self.time_events:
            self._ev.append([])
            self._mark()
        # The shard's search is ENQUEUED, the exchange step is enqueued right behind it on the same stream, and only then does
        # the host wait for the search (rdx_search_wait): the host-side cost of launching the collective overlaps the scan
        # instead of leaving the GPU idle after it. If some rank's search has to re-run overflowed queries (rare) every rank
        # learns it from the merged flags word and the exchange is repeated.
        deferred = hasattr(self.shard, "search_async") and self.shard.search_async(queries, k, s, r, c, f)
        if not deferred:
            self.shard.search(queries, k, s, r, c)
            f.zero_()                                 # a synchronous search leaves a complete partial
        self._mark()
        exchange = self.world > 1 or self.always_exchange
        if exchange:
            self._exchange(nq, k)
        self._open = (nq, k, deferred, exchange, queries)   # (the reference to `queries` keeps its memory from being recycled)

    def _exchange(self, nq: int, k: int) -> None:
        per_pad, local, allb, out = self._buffers(nq, k)
        self.exchanges += 1
        if self.host_staged:
            # rehearsal only (several ranks sharing ONE GPU over gloo, which cannot move device memory): same packed
            # layout, same merge kernel, the collective alone goes through host memory
            h_all = torch.empty(self.world * per_pad, dtype=torch.uint8)
            dist.all_gather_into_tensor(h_all, local.cpu(), group=self.group)
            allb.copy_(h_all)
        elif dist.is_initialized():
            dist.all_gather_into_tensor(allb, local, group=self.group)   # the ONE exchange step (RCCL over xGMI)
        else:
            allb.copy_(local)                          # always_exchange without a process group: one part, gathered by a copy
        self._mark()
        self.shard.merge_packed(allb, per_pad, self.world, nq, k, out[0], out[1], out[2])
        self._mark()

    def search_end(self):
        if self._open is None:
            raise RuntimeError("search_end() without search_begin()")
        nq, k, deferred, exchange, _keep = self._open
        self._open = None
        per_pad, local, allb, out = self._buffers(nq, k)
        if not exchange:
            if deferred:
                self.shard.search_wait()
            return self._views[(nq, k)][:3]
        mine = self.shard.search_wait() if deferred else False
        # every rank reads the same OR of the gathered flags words: all of them repeat the exchange, or none does
        if deferred and self.shard.merge_flag():
            if self.time_events and self._ev:
                self._ev[-1] = self._ev[-1][:2]    # the repeated exchange's events replace the first one's
            self._exchange(nq, k)   # some rank's fallback passes rewrote its partial after the first exchange: exchange and merge again
        elif mine:
            raise RuntimeError("internal: this rank re-ran queries but the merged flags word says nobody did")
        return out

    def breakdown(self, last: int = 0):
        """[scan_ms, exchange_ms, merge_ms] averaged over the (last N) steps recorded with time_events (call after a synchronise)"""
        evs = [e for e in self._ev if len(e) == 4]
        if last:
            evs = evs[-last:]
        if not evs:
            return None
        n = len(evs)
        return [sum(e[i].elapsed_time(e[i + 1]) for e in evs) / n for i in range(3)]
