#!/usr/bin/env python3
"""bench.py — whole-job queries/s of the cosine top-k hot path on N MI355X (contract: see the task statement).

A "step" is one pass of the hot path over one batch of synthetic queries already resident in HBM:
librdx search of this rank's corpus shard (K1 normalise -> MFMA scan with fused threshold top-k -> exact
re-score) + ONE RCCL all-gather of the packed partial top-k + the merge kernel. The corpus total is fixed
while N grows (strong scaling: BASELINE.json config "10M x 1024 row-sharded over 8 GPUs").

    python bench.py [--gpus N --steps K --warmup W] [--workload c4|c3|c2|c5|c1] [--rows R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line. Extra keys beyond the contract: roofline, cpu_baseline, recall_at_10, path_stats,
other_configs (c1 / c2 / c3 as short legs behind the timed region of the default c4 run).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from rag_dpo_amd import synth
from rag_dpo_amd.sharded import HipShard, ShardedSearcher, shard_range

# BASELINE.json configs. c1 = CNIL-sized synthetic corpus, the reference's own shape.
WORKLOADS = {
    "c1": dict(rows=16_919, dim=1024, batch=4, k=50, corpus="fp32",
               desc="CNIL-sized synthetic corpus 16,919 x 1024 fp32, 4 queries (one question's expansions), top-50"),
    "c2": dict(rows=100_000, dim=1024, batch=64, k=10, corpus="fp32",
               desc="synthetic 100k x 1024 fp32, batch=64 queries, top-10"),
    "c3": dict(rows=1_000_000, dim=1024, batch=256, k=100, corpus="fp32",
               desc="synthetic 1M x 1024 fp32, batch=256 queries, top-100"),
    "c4": dict(rows=10_000_000, dim=1024, batch=1024, k=10, corpus="fp32",
               desc="synthetic 10M x 1024 fp32 row-sharded over the GPUs, batch=1024 queries, top-10, all-gather merge"),
    "c5": dict(rows=10_000_000, dim=1024, batch=1024, k=10, corpus="bf16",
               desc="synthetic 10M x 1024 bf16 corpus row-sharded over the GPUs, batch=1024 queries, top-10, each step "
                    "encodes its 1024 query texts first (BGE-M3 architecture = XLM-R-large, random-init fp16, PyTorch-ROCm; "
                    "--no-encode times the search alone)"),
}
PEAK_HBM_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/f16 MFMA ~2.5 PF


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_shard(shard: HipShard, lo: int, hi: int, dim: int, corpus_kind: str, device, total_rows: int, shape: str = "iid"):
    """this rank's rows [lo, hi) of the deterministic synthetic corpus of SURVEY.md §8(d) (N(0,1) rows, 1 % exact
    duplicates), generated in HBM chunk by chunk. shape "embed": the embedding-like corpus of synth.torch_embedlike_chunk
    (a common mean direction, documents of contiguous near-duplicate chunks) instead"""
    shard.index.reserve(hi - lo)
    j0, j1 = lo // synth.CHUNK, (hi + synth.CHUNK - 1) // synth.CHUNK
    for j in range(j0, j1):
        r0 = j * synth.CHUNK
        if shape == "embed":
            rows = synth.torch_embedlike_chunk(j, min(synth.CHUNK, total_rows - r0), dim, device, total_rows)
        else:
            rows = synth.torch_corpus_chunk(j, min(synth.CHUNK, total_rows - r0), dim, device)
        a, b = max(lo, r0) - r0, min(hi, r0 + synth.CHUNK) - r0
        rows = rows[a:b].contiguous()
        if corpus_kind == "bf16":
            shard.index.add_bf16(rows.to(torch.bfloat16))
        else:
            shard.index.add(rows)
        del rows
    torch.cuda.synchronize(device)


def effective_cpus() -> int:
    """host cores this process may really use: min(os.cpu_count(), affinity mask, cgroup cpu.max quota)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_and_recall(shard: HipShard, queries: torch.Tensor, wl: dict, total_rows: int, planted=None):
    """rank 0, N=1 only. The oracle is the CHECKER and the timed CPU baseline; never the thing shipped."""
    from oracle import oracle as O
    O.build()
    dim, k, B = wl["dim"], wl["k"], wl["batch"]
    n = len(shard)
    sample = min(n, 262_144)
    rows_hat = shard.index.get(np.arange(sample, dtype=np.int64))        # stored (normalised) rows -> host
    q = queries.cpu().numpy()
    qhat = O.normalize_rows(q)
    # --- CPU baseline: fp32 BLAS sgemm + partial sort on all host cores, bounded sample, scaled to the full corpus
    threads = effective_cpus()
    bq = min(B, 1024)
    t_best, reps = None, 0
    t_end = time.time() + 12.0
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=threads):
        while reps < 3 or (time.time() < t_end and reps < 20):
            t0 = time.time()
            O.topk_blas_f32(rows_hat, qhat[:bq], min(k, sample))
            dt = time.time() - t0
            t_best = dt if t_best is None else min(t_best, dt)
            reps += 1
    qps_sample = bq / t_best
    cpu = {
        "value": round(qps_sample * sample / total_rows, 2), "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"oracle.topk_blas_f32 (numpy fp32 sgemm + argpartition, {threads} threads) on the first {sample} "
                  f"normalised rows x {bq} queries, best of {reps} ({t_best*1e3:.0f} ms); value = sample QPS x {sample}/{total_rows} rows",
    }
    # --- recall@k vs the exact C oracle on the same bounded sample (GPU restricted to it by the row bitmap)
    nchk = min(B, 64)
    es, er, ec = O.cosine_topk(rows_hat, q[:nchk], k)
    allow = np.zeros(n, dtype=bool)
    allow[:sample] = True
    gs, gr, gc = shard.index.search(q[:nchk], k, O.pack_mask(allow, n))
    hits = sum(len(set(er[b, :ec[b]].tolist()) & set(gr[b, :gc[b]].tolist())) for b in range(nchk))
    recall = hits / max(1, int(ec.sum()))
    exact_ids = bool((gr == er).all() and (gc == ec).all())
    max_ds = float(np.abs(gs[:, :k].astype(np.float64) - es.astype(np.float64)).max()) if k else 0.0
    # --- full-size properties: (1) every id the full search returned, re-scored by the oracle, equals the GPU score;
    # (2) the planted queries (q = c_i + 0.3 eps, §8d) find their row first (or an exact duplicate with a lower id);
    # (3) the exact full scan K5 (fp32, no MFMA, no threshold heuristics; itself oracle-checked at small N) over ALL
    # rows returns the same ids and score bits as the MFMA path for 16 queries: no better row exists anywhere
    fs, fr, fc = shard.index.search(q[:nchk], k)
    ok_full = True
    for b in range(nchk):
        rr = fr[b, :fc[b]]
        ref = O.scores(shard.index.get(rr), qhat[b])
        ok_full &= bool((ref == fs[b, :fc[b]]).all()) and bool((np.diff(fs[b, :fc[b]].astype(np.float64)) <= 0).all())
    planted_ok, n_pl = True, 0
    for qi, ri in planted or []:
        if qi < nchk and k > 0:
            n_pl += 1
            if fr[qi, 0] != ri:
                a, b_ = shard.index.get(np.array([fr[qi, 0], ri], dtype=np.int64))
                planted_ok &= bool(fr[qi, 0] < ri and (a == b_).all())
    nx = min(nchk, 16)
    shard.index.set_option("force_exact", 1)
    xs, xr, xc = shard.index.search(q[:nx], k)
    shard.index.set_option("force_exact", 0)
    exact_scan_equal = bool((xr == fr[:nx]).all() and (xs == fs[:nx]).all() and (xc == fc[:nx]).all())
    rec = {"value": round(recall, 6), "k": k, "queries": nchk, "sample_rows": sample, "ids_bit_exact": exact_ids,
           "max_abs_score_diff": max_ds, "full_corpus_returned_scores_match_oracle": ok_full,
           "planted_queries_checked": n_pl, "planted_rows_found_first": planted_ok,
           "full_corpus_exact_scan_queries": nx, "full_corpus_exact_scan_equals_mfma_path": exact_scan_equal}
    return cpu, rec


INFINITY_CACHE_BYTES = 256 * 2 ** 20


def _bytes_source(bytes_per_launch: float) -> str:
    """where a launch's algorithmic bytes come from when every step reads the same buffers again: a working set below the 256 MB
    Infinity Cache stays there between steps, and a fraction "of 8 TB/s" is then a yardstick, not an HBM measurement"""
    if bytes_per_launch <= INFINITY_CACHE_BYTES:
        return "Infinity-Cache resident between steps (working set <= 256 MB): frac is against the HBM yardstick, the bytes do not come from HBM"
    return "HBM (working set larger than the 256 MB Infinity Cache)"


def run_other_config(name: str, device, steps: int, warmup: int) -> dict:
    """One more BASELINE config in the same process, AFTER the timed region of the line's own workload (never inside it): its own
    index, `steps` timed searches with the corpus and the queries resident, the dominant kernel timed by librdx's HIP events on the
    search's stream, and the ids / score bits of a few queries compared with the C oracle (the whole corpus where that takes a
    second, otherwise the first 131 072 rows selected by bitmap). Same formulas as the main line's `roofline`."""
    from oracle import oracle as O
    O.build()
    wl = WORKLOADS[name]
    rows, dim, B, k = wl["rows"], wl["dim"], wl["batch"], wl["k"]
    t0 = time.time()
    sh = HipShard(dim, device.index or 0, row_offset=0)
    try:
        build_shard(sh, 0, rows, dim, wl["corpus"], device, rows)
        se = ShardedSearcher(sh)
        q = synth.torch_queries(B, dim, device, total_rows=rows)
        sh.index.set_option("profile", 2)
        for _ in range(warmup):
            se.search(q, k)
        torch.cuda.synchronize(device)
        scan_ms = exact_ms = 0.0
        t1 = time.perf_counter()
        for _ in range(steps):
            se.search(q, k)
            st = sh.index.last_stats_struct()
            scan_ms += st.ms_scan_main
            exact_ms += st.ms_exact
        torch.cuda.synchronize(device)
        ms_step = (time.perf_counter() - t1) / steps * 1e3
        stats = sh.index.last_stats()
        dim_pad = (dim + 63) // 64 * 64
        if stats["path"] == 0:
            kern_ms = scan_ms / steps
            flops, by = 2.0 * B * rows * dim_pad, rows * dim_pad * 2.0 + B * dim_pad * 2.0
            if flops / (PEAK_MFMA_TFLOPS * 1e12) >= by / (PEAK_HBM_GBS * 1e9):
                bound, frac = "mfma", flops / (kern_ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS
            else:
                bound, frac = "hbm", by / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
            kernel = "main scan (k_scan / k_scan_small)"
        else:
            kern_ms = exact_ms / steps
            by = -(-B // 4) * rows * dim * 4.0 + B * rows * 4.0 * 2
            bound, frac, kernel = "hbm", by / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "k_exact_scores + k_select_dense (exact path)"
        # the checker: ids and score bits against the C oracle
        nchk = min(B, 8)
        qh = q[:nchk].cpu().numpy()
        sample = rows if rows <= 131_072 else 131_072
        rows_hat = sh.index.get(np.arange(sample, dtype=np.int64))
        es, er, ec = O.cosine_topk(rows_hat, qh, k)
        allow = None
        if sample < rows:
            a = np.zeros(rows, dtype=bool)
            a[:sample] = True
            allow = O.pack_mask(a, rows)
        gs, gr, gc = sh.index.search(qh, k, allow)
        exact = bool((gr == er).all() and (gs == es).all() and (gc == ec).all())
        return {"workload": wl["desc"], "steps": steps, "ms_per_step": round(ms_step, 4), "queries_per_s": round(B / ms_step * 1e3, 1),
                "main_kernel": kernel, "main_kernel_ms": round(kern_ms, 4), "bound": bound, "frac": round(frac, 4),
                "bytes_source": _bytes_source(by),
                "ids_bit_exact_vs_oracle": exact, "oracle_check": f"{nchk} queries x {sample} rows" + ("" if sample == rows else " (bitmap-selected sample)"),
                "emitted_per_query": round(stats["emitted"] / max(1, B), 1), "retried_queries": stats["retried_queries"],
                "exact_fallback_queries": stats["exact_queries"], "build_and_run_s": round(time.time() - t0, 2)}
    finally:
        sh.index.close()


def self_launch(n: int) -> int:
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] launching", " ".join(cmd))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                       # rank 0 prints exactly one JSON line; anything else goes to stderr
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 100 for c4, 20 for c5; 200 for c3, 500 for c1 / c2, whose steps are "
                                                         "fractions of a millisecond: 20 of them end before the clocks have ramped)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps first (default: 20 for c4, 3 for c5; 20 / 50 for the small workloads)")
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override the workload's total corpus rows")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / recall leg")
    ap.add_argument("--no-encode", action="store_true", help="c5: leave the BGE-M3 query encode out of the step")
    ap.add_argument("--enc-buckets", type=int, default=4, help="c5: most length buckets (forwards) per encoded batch; 1 = one forward padded to the longest text")
    ap.add_argument("--enc-graphs", action="store_true", help="c5: replay the fused encoder forward as a HIP graph (one launch per encode)")
    ap.add_argument("--enc-torch-ops", action="store_true", help="c5: the packed forward on torch operations only (without librdx's attention and add + LayerNorm kernels)")
    ap.add_argument("--enc-module-forward", action="store_true", help="c5: the checkpoint's module-by-module forward over the padded batch instead of the packed forward")
    ap.add_argument("--no-overlap", action="store_true", help="c5: encode then search on one stream (no pipelining of batch i+1's encode with batch i's search)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=INT", help="developer: rdx_index_set_option before the run")
    ap.add_argument("--fp32-master", action="store_true", help="c5: keep the normalised fp32 rows as the exact copy (6 instead of 4 B/element)")
    ap.add_argument("--timer", choices=["events", "stamps"], default=None,
                    help="how the dominant kernel is timed inside the timed region: HIP events on its stream (default) or the kernel's own clock stamps")
    ap.add_argument("--profile-all", action="store_true", help="HIP events around every kernel of a search (path_stats.ms), not only the main scan")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1: initialise torch.distributed (nccl = RCCL) anyway and run the all-gather + merge with world 1 "
                         "(the exchange path on the real backend when only one GPU is there)")
    ap.add_argument("--check-merged", action="store_true",
                    help="N>1: rank 0 also builds the WHOLE corpus in one index and checks that the merged result is bit-identical "
                         "(the default cross-rank exactness leg needs no second copy of the corpus)")
    ap.add_argument("--corpus-shape", choices=["iid", "embed"], default="iid",
                    help="iid: SURVEY.md §8d's N(0,1) corpus (the judged line). embed: an embedding-like corpus — a common mean direction (unrelated rows "
                         "have cosine ~0.5) and 2000 documents of contiguous near-duplicate chunks, queries near documents (rag_dpo_amd/synth.py)")
    ap.add_argument("--no-others", action="store_true", help="c4, N=1: skip the short legs of configs c1 / c2 / c3 behind the timed region (other_configs)")
    ap.add_argument("--no-check", action="store_true", help="N>1 / --force-dist: skip the cross-rank exactness leg and the step breakdown")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a ONE-GPU box: RDX_BENCH_REHEARSAL=1 puts every rank on cuda:0 and runs the
    # collective over gloo through host memory. Never used by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("RDX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this parent has not touched the GPU (no HIP call so far) and never will;
        # it starts the N ranks as a CHILD torch.distributed.run (one process per GPU over RCCL), relays rank 0's JSON
        # line and exits with the child's code. Never an exec from a process that initialised the GPU.
        sys.exit(self_launch(args.gpus))
    # stdout carries exactly ONE line, the JSON: everything else this process (or a library inside it: RCCL prints its
    # version banner on stdout) writes to file descriptor 1 goes to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1 or args.force_dist:
        if args.force_dist and "RANK" not in os.environ:     # plain `python bench.py --force-dist`: a one-rank group in this process
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # nccl == RCCL on ROCm

    if args.steps <= 0:
        args.steps = {"c1": 500, "c2": 500, "c3": 200, "c4": 100}.get(args.workload, 20)    # (>= 100 timed batches behind
    if args.warmup < 0:                                                                       #  >= 20 warm-up ones; the driver passes its own)
        args.warmup = {"c1": 50, "c2": 50, "c3": 20, "c4": 20}.get(args.workload, 3)
    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        wl["rows"] = args.rows
    if args.batch:
        wl["batch"] = args.batch
    if args.k:
        wl["k"] = args.k
    rows, dim, B, k = wl["rows"], wl["dim"], wl["batch"], wl["k"]
    lo, hi = shard_range(rows, world, rank)

    t_build = time.time()
    shard = HipShard(dim, local_rank, row_offset=lo)
    if wl["corpus"] == "bf16" and not args.fp32_master:
        shard.index.set_option("compact_master", 1)   # config 5: raw bf16 rows + divisors as the exact copy: 4 B/element in HBM
    build_shard(shard, lo, hi, dim, wl["corpus"], device, rows, shape=args.corpus_shape)
    log(f"[rank {rank}] shard rows [{lo}, {hi}) resident in {time.time() - t_build:.1f}s")
    searcher = ShardedSearcher(shard, host_staged=rehearsal, always_exchange=args.force_dist)
    if args.corpus_shape == "embed":
        queries, planted = synth.torch_embedlike_queries(B, dim, device)[0], None
    else:
        queries, planted = synth.torch_queries(B, dim, device, total_rows=rows, return_planted=True)   # 10 % planted (§8d)
    # ONE query batch for all ranks: rank 0's, broadcast once, outside the timed region (the rank that took the
    # request hands it to the others). Nothing rests on every rank's generator producing the same bits.
    if world > 1 or args.force_dist:
        searcher.broadcast_queries(queries, 0)
    # HIP events on the stream the kernels run on: 2 = around the dominant kernel (the main scan) only, which is what the
    # timed region carries; --profile-all records every kernel boundary (7 events per search: visible in small configs)
    # --timer stamps (profile = 3): no events at all, the dominant kernel stamps its own first-workgroup start and last-workgroup
    # end (100 MHz clock). Measures what the two event records cost a small search (c1 / c2 / c3: 4 / 3 / 14 us per step) and the
    # kernel without its launch ramp; the default stays HIP events, as the bench contract asks.
    timer = args.timer or "events"
    shard.index.set_option("profile", 1 if args.profile_all else (2 if timer == "events" else 3))
    for kv in args.set:
        shard.index.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # C5: every step first encodes its B query texts (every rank encodes the same texts with the same weights: the query
    # batch is replicated) and hands the device tensor straight to the search
    encode = args.workload == "c5" and not args.no_encode
    provider, texts, enc_ev = None, None, []
    if encode:
        from rag_dpo_amd.embedding_provider import EmbeddingProvider
        provider = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device=str(device), dtype=torch.float16,
                                     batch_size=int(os.environ.get("RDX_ENC_BATCH", "1024")))
        provider.packed_forward = not args.enc_module_forward
        if args.enc_torch_ops:
            provider.fused_kernels = False
        provider.encoder_graphs = True if args.enc_graphs else None
        provider.load()
        provider.max_buckets = max(1, args.enc_buckets)
        texts = synth.query_texts(B)

    def step():
        if not encode:
            return searcher.search(queries, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        q = provider.embed_device(texts)
        e1.record()
        enc_ev.append((e0, e1))
        return searcher.search(q, k)

    for _ in range(args.warmup):
        step()
    barrier()
    enc_ev.clear()
    scan_ms, tot_ms, stats = 0.0, 0.0, None
    exact_ms = 0.0
    serial = None
    host_ms = [0.0, 0.0, 0.0]
    if encode and not args.no_overlap:
        # C5 step = encode B texts + search, software-pipelined on ONE stream: the search of batch i is enqueued
        # (search_begin), the encode of batch i+1 — tokenising on the host, then the forward's launches — is issued behind it
        # on the same stream, and only then does the host wait for the search (search_end). Both phases are compute-bound on
        # the same CUs, so they do not run beside each other on the GPU (an earlier version on two streams with a worker thread
        # measured SLOWER than the serial order once the tokeniser stopped being the slow part); what the pipeline hides is
        # the host's share of the encode (tokenise + ~500 launches) behind the 16 ms scan. K steps are still K encodes + K
        # searches inside the timed region. A short serial leg (encode, wait, search, wait) is reported beside it.
        n_serial = max(2, min(5, args.steps))
        t0s = time.perf_counter()
        for _ in range(n_serial):
            step()
        torch.cuda.synchronize(device)
        serial = {"steps": n_serial, "ms_per_step": round((time.perf_counter() - t0s) / n_serial * 1e3, 4),
                  "encode_avg_ms": round(sum(a.elapsed_time(b) for a, b in enc_ev) / max(1, len(enc_ev)), 3)}
        enc_ev.clear()
        barrier()
        t0 = time.perf_counter()

        def encode_now():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            q_ = provider.embed_device(texts)
            e1.record()
            enc_ev.append((e0, e1))
            return q_

        q_next = encode_now()
        host_ms = [0.0, 0.0, 0.0]          # where the HOST spends a pipelined step: search_begin, the next encode's enqueue, search_end (its wait)
        for i in range(args.steps):
            q_cur = q_next
            h0 = time.perf_counter()
            searcher.search_begin(q_cur, k)
            h1 = time.perf_counter()
            if i + 1 < args.steps:
                q_next = encode_now()
            h2 = time.perf_counter()
            searcher.search_end()
            h3 = time.perf_counter()
            for j, d in enumerate((h1 - h0, h2 - h1, h3 - h2)):
                host_ms[j] += d * 1e3
            st = shard.index.last_stats()
            scan_ms += st["ms_scan_main"]
            exact_ms += st["ms_exact"]
            tot_ms += st["ms_total"]
            stats = st
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            st = shard.index.last_stats_struct()   # host struct copy, no device work (three field reads: a dict of all 20 fields
            scan_ms += st.ms_scan_main             #  cost 5 us per step, 8 % of a c1 step)
            exact_ms += st.ms_exact
            tot_ms += st.ms_total
        stats = shard.index.last_stats()
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1 (and the one-rank RCCL group of --force-dist): the line proves itself ---------------------------------
    # (1) cross-rank exactness, no oracle over 10 M rows needed: for 32 queries every rank ALSO answers from its shard with the
    #     exact full scan K5 (fp32, no MFMA, no thresholds; oracle-checked at small N by the test suite), those partials go
    #     through the SAME all-gather + merge, and the merged lists must equal the MFMA path's merged lists bit for bit: no
    #     better row exists in any shard, and the exchange + merge handled both identically. (2) every rank's merged output
    #     is checksummed and the checksums compared: all ranks hold the same answer. (3) what RCCL ran on. (4) where a step's
    #     time goes on rank 0 (events around the pieces, in a separate short leg: never inside the timed region).
    dist_check, rccl_info, step_breakdown = None, None, None
    try:
        if (world > 1 or args.force_dist):
            props = torch.cuda.get_device_properties(device)
            mine = {"rank": rank, "device_index": local_rank, "name": props.name, "gcnArch": getattr(props, "gcnArchName", "?").split(":")[0]}
            infos = [None] * world
            dist.all_gather_object(infos, mine)
            try:
                ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if not rehearsal else None
            except Exception:
                ver = None
            rccl_info = {"world": world, "backend": dist.get_backend(), "rccl_version": ver, "devices": infos,
                         "collectives_per_step": "1 all_gather_into_tensor of the packed partials (rows|scores|counts|flags), nothing else",
                         "queries": "rank 0's batch, broadcast once before the timed region"}
            if not args.no_check:
                nchk = min(B, 32)
                qs = queries[:nchk].contiguous()
                m_s, m_r, m_c = [t.clone() for t in searcher.search(qs, k)]
                full_s, full_r, full_c = [t.clone() for t in searcher.search(queries, k)]     # the whole batch, for the checksum
                shard.index.set_option("force_exact", 1)
                x_s, x_r, x_c = [t.clone() for t in searcher.search(qs, k)]
                shard.index.set_option("force_exact", 0)
                torch.cuda.synchronize(device)
                exact_equal = bool((x_r == m_r).all() and (x_s == m_s).all() and (x_c == m_c).all())
                prefix_equal = bool((full_r[:nchk] == m_r).all() and (full_s[:nchk] == m_s).all())   # a query's answer does not depend on its batch
                # checksum of everything this rank would hand to a caller: rows, score BITS, counts (wrapping int64 sums + a weighted one)
                w = torch.arange(1, full_r.numel() + 1, device=full_r.device, dtype=torch.int64)
                bits = full_s.view(torch.int32).to(torch.int64)
                ck = torch.stack([full_r.sum(), (full_r.flatten() * w).sum(), bits.sum(), (bits.flatten() * w).sum(),
                                  full_c.to(torch.int64).sum(), torch.tensor(int(exact_equal and prefix_equal), device=full_r.device)])
                ck_h = ck.cpu() if rehearsal else ck
                allck = torch.empty(world * ck.numel(), dtype=torch.int64, device=ck_h.device)   # (flat: gloo refuses a 2-D output)
                dist.all_gather_into_tensor(allck, ck_h)
                allck = allck.cpu().view(world, ck.numel())
                dist_check = {"queries_exact_leg": nchk,
                              "per_shard_exact_scan_merged_equals_mfma_merged": bool(allck[:, 5].all().item()),
                              "merged_identical_on_all_ranks": bool((allck[:, :5] == allck[0, :5]).all().item()),
                              "checksum_rank0": [int(v) for v in allck[0, :5]],
                              "exchanges_per_search": "1 (a second one only when the merged flags word says a rank re-ran overflowed queries)"}
                # step breakdown on rank 0: search kernels / all-gather / merge, events on the stream, min(steps, 10) extra steps
                searcher.time_events = True
                nb = max(3, min(args.steps, 10))
                ex0 = searcher.exchanges
                for _ in range(nb):
                    searcher.search(queries, k)
                barrier()
                bd = searcher.breakdown(last=nb)
                searcher.time_events = False
                if bd is not None:
                    step_breakdown = {"steps": nb, "scan_ms": round(bd[0], 4), "exchange_ms": round(bd[1], 4), "merge_ms": round(bd[2], 4),
                                      "exchanges": searcher.exchanges - ex0,
                                      "note": "rank 0, torch events on the search's stream: everything librdx enqueues for the shard's search "
                                              "(K1 .. k_finish) | all_gather_into_tensor | rdx_merge_topk_packed; a separate leg after the timed region"}
    except Exception as e:   # the checker legs must never take the measured number down with them (every rank runs the same code
        log(f"[rank {rank}] distributed check leg failed: {e!r}")   # on the same data: a failure here is the same failure on every rank)
        dist_check = {"error": repr(e)}

    merged_ok = None
    if args.check_merged and world > 1:
        ms_, mr_, mc_ = [t.clone() for t in searcher.search(queries, k)]   # identical on every rank by construction
        if rank == 0:
            from rag_dpo_amd.engine import HipIndex
            whole = HipShard(dim, local_rank, row_offset=0)
            build_shard(whole, 0, rows, dim, wl["corpus"], device, rows)
            ws = torch.empty_like(ms_); wr = torch.empty_like(mr_); wc = torch.empty_like(mc_)
            whole.search(queries, k, ws, wr, wc)
            torch.cuda.synchronize(device)
            merged_ok = bool((wr == mr_).all() and (ws == ms_).all() and (wc == mc_).all())
            whole.index.close()
    enc_stats = None
    if encode and rank == 0:
        # one more encode, outside the timed region, with events around every bucket's forward
        provider.time_buckets = True
        provider.embed_device(texts)
        provider.time_buckets = False
        es = provider.last_encode_stats
        enc_stats = {"tokens_real": es["tokens_real"], "tokens_padded": es["tokens_padded"],
                     "tokens_real_over_padded": es["real_over_padded"],
                     "tokens_padded_if_one_forward": es["tokens_padded_one_width"], "buckets": es["buckets"],
                     "forward": ("the checkpoint's module forward over the padded batch" if provider._packed is None else
                                 "packed + librdx kernels: every layer over the real tokens only (no padding anywhere), one GEMM for Q/K/V, rdx_enc_attention_f16, "
                                 "rdx_enc_add_layernorm_f16; GEMMs and GELU are torch's" if provider._packed.fused else
                                 "packed, torch operations: token-wise layers over the real tokens only, padding only around the attention, one GEMM for Q/K/V"),
                     "rule": "token-count-sorted rows cut into <= %d buckets (multiples of %d rows), each forwarded at its own width" % (provider.max_buckets, provider.bucket_granule)}
    out = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        qps = B * args.steps / elapsed
        n_local = hi - lo
        dim_pad = (dim + 63) // 64 * 64
        launch_ms = scan_ms / args.steps
        flops = 2.0 * B * n_local * dim_pad          # algorithmic flop of one main-scan launch
        # Two byte counts, both stated: what the launch READS by construction — the fp16 scan copy once +
        # the fp16 query images — and SURVEY.md §8(d)'s figure for the batch, N_local*d*s + B*d*4 + B*k*8 with s = 4
        # (fp32 corpus; 2 for the bf16 corpus of config 5): the bytes a scan of the corpus AS DELIVERED would move.
        bytes_ = n_local * dim_pad * 2.0 + B * dim_pad * 2.0
        s_in = 2.0 if wl["corpus"] == "bf16" else 4.0
        bytes_8d = n_local * dim * s_in + B * dim * 4.0 + B * k * 8.0
        t_hbm, t_mfma = bytes_ / (PEAK_HBM_GBS * 1e9), flops / (PEAK_MFMA_TFLOPS * 1e12)
        roof = None
        if stats and stats["path"] == 0 and launch_ms > 0:
            traffic, traffic_src = None, None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp))
                    ent = tj.get(f"{args.workload}_n{world}")
                    if isinstance(ent, dict):
                        traffic, traffic_src = ent.get("bytes_per_launch"), ent.get("source")
                    elif ent is not None:
                        traffic, traffic_src = ent, tj.get("_source")
                except Exception:
                    traffic = None
            sec = launch_ms * 1e-3
            if t_mfma >= t_hbm:
                ach = flops / sec / 1e12
                roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": traffic}
            else:
                ach = bytes_ / sec / 1e9
                roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic}
            roof.update({"kernel": "rdx::k_scan<BN,EPI_EMIT> (main scan)", "avg_launch_ms": round(launch_ms, 4),
                         "timer": ("HIP events recorded by librdx on the stream the kernel is launched on (option profile=2), "
                                   "averaged over the timed steps of THIS run") if (args.profile_all or timer == "events") else
                                  ("the kernel's own stamps (100 MHz wall clock): first workgroup start -> last workgroup end, option "
                                   "profile=3, averaged over the timed steps of THIS run; no event records on the stream"),
                         "launch_rows": n_local, "launch_queries": B, "flops_per_launch": flops,
                         "bytes_read_per_launch": bytes_, "bytes_read_is": "fp16 scan copy once + fp16 query images (what the kernel loads)",
                         "bytes_source": _bytes_source(bytes_),
                         "bytes_survey_8d_per_launch": bytes_8d, "bytes_survey_8d_is": f"N_local*d*{int(s_in)} + B*d*4 + B*k*8 (SURVEY.md §8d)",
                         "hbm_frac_of_8TBs": round(bytes_ / sec / 1e9 / PEAK_HBM_GBS, 4),
                         "hbm_frac_of_8TBs_on_survey_8d_bytes": round(bytes_8d / sec / 1e9 / PEAK_HBM_GBS, 4),
                         "mfma_frac_of_2.5PF": round(flops / sec / 1e12 / PEAK_MFMA_TFLOPS, 4),
                         "traffic_is": "fabric bytes per launch of this kernel from an EARLIER rocprofv3 --pmc run "
                                       "(2*FETCH_SIZE + WRITE_SIZE, separate passes), not measured in this run; at B > 256 the part "
                                       "above the algorithmic bytes is query-image lines re-fetched from the Infinity Cache, the corpus "
                                       "crosses the fabric once (profiles/r02/traffic_split_4Mrows_pmc.txt)",
                         "traffic_source": traffic_src})
        if stats and stats["path"] == 1 and exact_ms > 0:
            # small corpora are answered by the exact full scan alone (K5a k_exact_scores + K5b k_select_dense): one fp32 read
            # of the corpus per <= 4 queries, HBM/latency-bound
            sec = exact_ms / args.steps * 1e-3
            by = -(-B // 4) * n_local * dim * 4.0 + B * n_local * 4.0 * 2
            roof = {"bound": "hbm", "achieved": round(by / sec / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(by / sec / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                    "kernel": "rdx::k_exact_scores + rdx::k_select_dense (exact path of small corpora)",
                    "avg_launch_ms": round(exact_ms / args.steps, 4), "timer": ("HIP events recorded by librdx around the two kernels" if (args.profile_all or timer == "events") else
                                                                               "the kernels' own stamps (100 MHz wall clock): first block of the scoring kernel -> last block of the select kernel; no event records on the stream"),
                    "bytes_read_per_launch": by,
                    "bytes_read_is": "ceil(B/4) fp32 passes over the corpus + the dense score rows written and read back",
                    "launch_rows": n_local, "launch_queries": B}
        # the same corpus swept with a small batch: the HBM-bound regime of the same kernel (BASELINE.json's
        # ">= 50 % of the HBM roofline on the 10M x 1024 scan" is about THIS regime; at B = 1024 the scan is MFMA-bound)
        small = None
        if world == 1 and stats and stats["path"] == 0:
            try:
                bs = 64
                qs = synth.torch_queries(bs, dim, device, total_rows=rows) if args.corpus_shape == "iid" else synth.torch_embedlike_queries(bs, dim, device)[0]
                for _ in range(2):
                    searcher.search(qs, k)
                torch.cuda.synchronize(device)
                ms_s, n_s, t0s = 0.0, 10, time.perf_counter()
                for _ in range(n_s):
                    searcher.search(qs, k)
                    ms_s += shard.index.last_stats()["ms_scan_main"]
                torch.cuda.synchronize(device)
                wall = (time.perf_counter() - t0s) / n_s
                by = n_local * dim_pad * 2.0 + bs * dim_pad * 2.0
                gbs = by / (ms_s / n_s * 1e-3) / 1e9
                small = {"batch": bs, "k": k, "queries_per_s": round(bs / wall, 1), "avg_launch_ms": round(ms_s / n_s, 4),
                         "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4)}
            except Exception as e:
                log(f"small-batch leg failed: {e!r}")
        # "queries already on device (H2D of B x 4 KB reported separately)". Measured here, never part of
        # `value`: the H2D copy of the fp32 query batch from pinned host memory, and the whole search called with HOST
        # pointers (rdx_search RDX_HOST: pageable numpy in, H2D, search, D2H of the k results, numpy out)
        pcie = None
        if world == 1:
            try:
                hq = queries.cpu().pin_memory()
                dq = torch.empty_like(queries)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dq.copy_(hq, non_blocking=True)
                torch.cuda.synchronize(device)
                e0.record()
                for _ in range(10):
                    dq.copy_(hq, non_blocking=True)
                e1.record()
                torch.cuda.synchronize(device)
                h2d_ms = e0.elapsed_time(e1) / 10
                qn = queries.cpu().numpy()
                for _ in range(2):
                    shard.index.search(qn, k)
                n_h = max(3, min(args.steps, 10))
                t0h = time.perf_counter()
                for _ in range(n_h):
                    shard.index.search(qn, k)
                wall_h = (time.perf_counter() - t0h) / n_h
                pcie = {"h2d_queries_ms": round(h2d_ms, 4), "h2d_bytes": int(B * dim * 4),
                        "h2d_GBps": round(B * dim * 4 / (h2d_ms * 1e-3) / 1e9, 1) if h2d_ms > 0 else None,
                        "host_pointer_search_ms": round(wall_h * 1e3, 4), "host_pointer_queries_per_s": round(B / wall_h, 1),
                        "note": "PCIe-inclusive rate (host numpy in -> host numpy out through rdx_search RDX_HOST); reported, never `value`"}
            except Exception as e:
                log(f"pcie leg failed: {e!r}")
        # the other single-GPU BASELINE configs, short legs in the same process AFTER everything timed above
        others = None
        if world == 1 and args.workload == "c4" and args.corpus_shape == "iid" and not args.no_others and not args.rows and not args.batch and not args.k:
            others = {}
            for name, n_steps, n_warm in (("c2", 500, 50), ("c3", 200, 20), ("c1", 500, 50)):
                try:
                    others[name] = run_other_config(name, device, n_steps, n_warm)
                except Exception as e:   # a failing extra leg must not take the line down
                    log(f"other_configs leg {name} failed: {e!r}")
                    others[name] = {"error": repr(e)}
        cpu, rec = None, None
        if world == 1 and not args.no_cpu:
            try:
                cpu, rec = cpu_baseline_and_recall(shard, queries, wl, rows, planted)
            except Exception as e:   # the checker must never take the measured number down with it
                log(f"cpu_baseline/recall leg failed: {e!r}")
        out = {
            "metric": "queries/sec, d=1024 cosine top-k (exact, ids == CPU oracle)", "value": round(qps, 1), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16 MFMA scan + f64 exact re-score",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "rows_total": rows, "rows_per_gpu": n_local, "dim": dim,
                       "batch": B, "k": k, "corpus_dtype": wl["corpus"],
                       "hbm_bytes_per_element": (4 if (wl["corpus"] == "bf16" and not args.fp32_master) else 6),
                       "synthetic_inputs": ("SURVEY.md §8d: N(0,1) rows with 1 % exact duplicate rows, N(0,1) queries with 10 % planted next to a row" if args.corpus_shape == "iid" else
                                            "embedding-like: row = mu + d_doc + sigma_doc * eps (unrelated rows: cosine ~0.5), 2000 documents of contiguous chunks with "
                                            "sigma_doc in [0.2, 0.6], every query near one document (rag_dpo_amd/synth.py torch_embedlike_chunk)"),
                       "parallelism": f"row-shard x{world} + all-gather merge"},
            "roofline": roof, "roofline_small_batch": small, "pcie_inclusive": pcie, "cpu_baseline": cpu, "recall_at_10": rec,
            "other_configs": others, "merged_equals_single_index": merged_ok, "rccl": rccl_info, "distributed_check": dist_check, "step_breakdown": step_breakdown,
            "encode": ({"model": "XLM-R-large (BGE-M3 architecture), random-init fp16, hashing tokenizer", "texts_per_step": B,
                        "avg_ms": round(sum(a.elapsed_time(b) for a, b in enc_ev) / max(1, len(enc_ev)), 3),
                        "pipelined_with_search": bool(serial is not None), "serial_leg": serial,
                        "host_ms_per_step": ({"search_begin": round(host_ms[0] / args.steps, 3), "next_encode_enqueue": round(host_ms[1] / args.steps, 3),
                                              "search_end_wait": round(host_ms[2] / args.steps, 3)} if serial is not None else None), "length_buckets": enc_stats,
                        "note": "encoder value parity unpinned (no BGE-M3 weights offline); the GEMMs and GELU are PyTorch-ROCm plumbing, the attention and the add + LayerNorm pairs "
                                "are librdx kernels when length_buckets.forward says so (each checked against a torch fp32 reference of the same op in tests/)"} if encode else None),
            "path_stats": {"avg_search_ms_events": round(tot_ms / args.steps, 4),"exact_fallback_queries": stats["exact_queries"],
                           "emitted_per_query": round(stats["emitted"] / max(1, B), 1),
                           "rescored_per_query": round(stats["rescored"] / max(1, B), 2), "sample_rows": stats["sample_rows"],
                           "tau_rank": int(stats["tau_rank"]), "retried_queries": stats["retried_queries"],
                           "xcd_finish_spread_ms": round(stats["xcd_finish_spread_ms"], 4),
                           "xcd_share_min_max": [round(stats["xcd_share_min"], 3), round(stats["xcd_share_max"], 3)],
                           "ms": {n_: round(stats[n_], 4) for n_ in ("ms_normalize", "ms_scan_sample", "ms_tau", "ms_scan_main",
                                                                     "ms_refine", "ms_exact")}},
        }
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
