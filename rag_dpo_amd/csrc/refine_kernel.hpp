// refine_kernel.hpp — K4: exact re-score + final ordering of the scan's candidates, and C1's merge of per-shard
// partial results after the RCCL all-gather.
#pragma once
#include "k_rows.hpp"

namespace rdx {

constexpr int REFINE_PMAX = 1024;    // most candidates re-scored exactly per query; more -> exact full scan
constexpr int REFINE_LIST = 7168;    // most scan hits gathered per query (56 KiB of LDS: with the 22 KiB of static LDS TWO blocks fit a CU's 160 KiB —
                                     // at B = 1024 the kernel runs in two rounds instead of four); more -> second pass / exact full scan
constexpr int REFINE_STREAMS = 512;  // most (query, stream) segments

struct RefineCounters {   // one per index, zeroed before every search
    unsigned long long emitted;
    unsigned long long rescored;
    int n_exact;          // queries handed to the exact full scan
    int bad;              // set by K1 when a query embedding holds NaN/Inf
    int oob;              // RDX_CHECK_BOUNDS builds: the scan computed a corpus address outside the scan copy
    int spec_fail;        // queries whose speculative threshold could not be verified (k_refine; they take the fallback passes)
    // option profile = 3 (kernels stamp their own times, no HIP events on the stream): when the first block of the exact path's
    // scoring kernel started (kept as max(~clock): zero = unset) and when the last block of its select kernel ended (100 MHz ticks)
    unsigned long long t_first_inv, t_last;
    int done;             // blocks of the search's last kernel that have finished (the last one runs the end-of-search work)
    int pad1;
};

// What the LAST kernel of a search leaves in pinned host memory (written straight over PCIe, no memcpy, no interrupt):
// the host spins on `seq` instead of paying a D2H copy plus hipStreamSynchronize (~25 us of a 0.1 ms search).
struct Mailbox {
    unsigned long long seq;            // written last, system-scope release: search number `seq` is complete
    unsigned long long emitted, rescored;
    int n_exact, bad;
    int oob, spec_fail;
    unsigned long long t_first, t_last;   // profile = 3, exact path (see RefineCounters)
    unsigned long long wg_times[1024]; // [grid][2] start/end stamps of the main scan's workgroups (XCD balancing)
};

// K6. End of every search: (1) small results of HOST callers go from the device result buffers to pinned host staging,
// (2) the counters (and the workgroup stamps) go to the mailbox, (3) the counter block is zeroed for the next search
// (searches on one index are serialised by its mutex and each one has completed before the next is enqueued),
// (4) the sequence number is published. The volumes are KBs: ONE block does it — since round 3 the last block to finish of the
// search's last kernel (k_refine / k_select_dense: finish_if_last below), which saves a launch and a gap on every search; the
// stand-alone kernel k_finish remains for option "fuse_finish" = 0.
struct FinishArgs {
    RefineCounters* ctr;     // NULL: this launch does not end a search
    Mailbox* mb;
    unsigned long long seq;
    const unsigned long long* wgt;
    int n_wgt;
    const uint32_t* s0; uint32_t* d0; int64_t w0;
    const uint32_t* s1; uint32_t* d1; int64_t w1;
    const uint32_t* s2; uint32_t* d2; int64_t w2;
    int32_t* out_flags;
    int may_redo;
};

__device__ __forceinline__ void finish_body(const FinishArgs& f) {   // all threads of ONE block
    RefineCounters* ctr = f.ctr;
    Mailbox* mb = f.mb;
    for (int64_t i = threadIdx.x; i < f.w0; i += blockDim.x) f.d0[i] = f.s0[i];
    for (int64_t i = threadIdx.x; i < f.w1; i += blockDim.x) f.d1[i] = f.s1[i];
    for (int64_t i = threadIdx.x; i < f.w2; i += blockDim.x) f.d2[i] = f.s2[i];
    for (int i = threadIdx.x; i < f.n_wgt; i += blockDim.x) mb->wg_times[i] = f.wgt[i];
    // every storing wave waits for its own stores, the barrier collects the waves, and ONE lane makes the block's stores visible to
    // the host (system-scope release) before it publishes the sequence number — not a system-scope fence in all 1024 threads
    // (MI355X_MICROARCH.md, inter-workgroup visibility: "every storing wave's s_waitcnt vmcnt(0) -> __syncthreads() -> lane-0 fence")
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // rdx_search_async(out_flags): "this partial is incomplete, the host half will redo some queries" — the word travels with
        // the packed partial through the all-gather (include/rdx.h); the same condition complete_chunk reports as *redone
        if (f.out_flags) {
            f.out_flags[0] = (f.may_redo && ctr->n_exact > 0 && !ctr->bad) ? 1 : 0;
            f.out_flags[1] = 0;
            f.out_flags[2] = 0;
            f.out_flags[3] = 0;
        }
        mb->emitted = ctr->emitted;
        mb->rescored = ctr->rescored;
        mb->n_exact = ctr->n_exact;
        mb->bad = ctr->bad;
        mb->oob = ctr->oob;
        mb->spec_fail = ctr->spec_fail;
        ctr->spec_fail = 0;
        mb->t_first = ~ctr->t_first_inv;
        mb->t_last = ctr->t_last;
        ctr->t_first_inv = 0;
        ctr->t_last = 0;
        ctr->oob = 0;
        ctr->emitted = 0;
        ctr->rescored = 0;
        ctr->n_exact = 0;
        ctr->bad = 0;
        ctr->done = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the block's stores (L2 written back) before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (always, behind a release fence: the compiler may drop its own wait)
        __hip_atomic_store(&mb->seq, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(1024) void k_finish(const FinishArgs f) { finish_body(f); }

// Tail of the search's last kernel (all threads of every block call it, at a point every thread reaches): the block's result
// stores are drained and released at agent scope by one lane, which takes a ticket; the block holding the last ticket acquires
// (its CU's L1 may hold stale lines of what the other blocks wrote) and runs the end-of-search work.
// (MI355X_MICROARCH.md, hand-off forms: storing waves' vmcnt(0) -> barrier -> one lane's release + agent-scope atomic add; the
//  workgroup whose add came last loads after its acquire and a barrier.)
__device__ __forceinline__ void finish_if_last(const FinishArgs& f) {
    if (!f.ctr) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int t = __hip_atomic_fetch_add(&f.ctr->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (int)gridDim.x - 1;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_last = last;
    }
    __syncthreads();
    if (s_last) finish_body(f);
}

// One block per query (256 threads, 1024 when k is large: the exact re-score runs one wave per candidate row).
//   The scan left, per (query, stream), cntw hits in a segment of capw slots: (coarse score, row) of every allowed row
//   whose coarse score >= tau[q]. They are gathered into LDS; c_k = k-th largest coarse score. Every true top-k row has
//   coarse >= c_k - 2E (|coarse-exact| <= E and k rows reach coarse c_k, hence exact c_k - E), so
//   P = {coarse >= c_k - 2E} contains the exact top-k; P is re-scored exactly from the fp32 master copy (fp64 lane-order
//   sum, oracle/rdx_oracle.c) and ranked (score desc, row asc).
//   A segment, list or P overflow cannot be answered here: the query is flagged for the exact full scan.
#ifdef RDX_REFINE_STAMPS   // developer build (tools/refine_stamps.py): where k_refine spends its time (block 0's phases, 100 MHz wall clock)
__device__ unsigned long long g_refine_stamps[16];
#define RDX_RSTAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_refine_stamps[i] = wall_clock64(); } while (0)
#else
#define RDX_RSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void refine_query(const uint2* __restrict__ cand, const uint32_t* __restrict__ cntw,
                                                int n_streams, uint32_t capw, uint32_t list_cap, int k, float two_e,
                                                const float* __restrict__ qhat, MasterView master, int dim,
                                                int64_t row_base, const int64_t* __restrict__ row_map,
                                                float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                int32_t* __restrict__ out_count, int32_t* __restrict__ exact_list,
                                                RefineCounters* __restrict__ ctr, const float* __restrict__ tau, float inv_scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* list = reinterpret_cast<uint2*>(smem);                        // [REFINE_LIST]
    __shared__ __attribute__((aligned(16))) uint32_t hist[HIST_WORDS];
    __shared__ uint32_t bc[4];
    __shared__ uint32_t seg_off[REFINE_STREAMS + 1];
    __shared__ float s_s[REFINE_PMAX];
    __shared__ int64_t s_r[REFINE_PMAX];
    __shared__ int n_p, overflow;
    const int q = blockIdx.x;
    float* o_s = out_score + (int64_t)q * k;
    int64_t* o_r = out_row + (int64_t)q * k;
    if (threadIdx.x == 0) overflow = 0;
    __syncthreads();
    RDX_RSTAMP(0);
    // segment sizes -> exclusive prefix. Wave 0 scans them 64 at a time with shuffles (n_streams <= 512: at most 8 rounds; the
    // serial loop this replaces was 3-5 us of the kernel at 256 streams)
    for (int w = threadIdx.x; w < n_streams; w += blockDim.x) {
        const uint32_t c = cntw[(int64_t)q * n_streams + w];
        if (c > capw) overflow = 1;
        seg_off[w + 1] = c > capw ? capw : c;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        uint32_t carry = 0;
        for (int base = 0; base < n_streams; base += 64) {
            const int w = base + l;
            uint32_t v = w < n_streams ? seg_off[w + 1] : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(v, off, 64);
                if (l >= off) v += t;
            }
            if (w < n_streams) seg_off[w + 1] = carry + v;      // inclusive prefix -> offset of segment w + 1
            carry += __shfl(v, 63, 64);
        }
        if (l == 0) seg_off[0] = 0;
    }
    __syncthreads();
    const uint32_t m = seg_off[n_streams];
    RDX_RSTAMP(1);
    if (threadIdx.x == 0) atomicAdd(&ctr->emitted, (unsigned long long)m);
    if (overflow || m > list_cap) {
        if (threadIdx.x == 0) exact_list[atomicAdd(&ctr->n_exact, 1)] = q;
        return;
    }
    if (m == 0 || k == 0) {
        rank_and_write(s_s, s_r, 0, k, o_s, o_r, out_count + q);
        return;
    }
    // gather: thread w walks segment w (independent loads across threads)
    for (int w = threadIdx.x; w < n_streams; w += blockDim.x) {
        const uint32_t a = seg_off[w], b = seg_off[w + 1];
        const uint2* seg = cand + ((int64_t)q * n_streams + w) * capw;
        for (uint32_t i = a; i < b; ++i) list[i] = seg[i - a];
    }
    __syncthreads();
    RDX_RSTAMP(2);
    const int64_t kk = (uint32_t)k < m ? k : m;
    int64_t n_gt;
    const uint32_t kth = block_kth_largest([&](int64_t i) { return f2key(__uint_as_float(list[i].x)); }, m, kk, hist, bc, &n_gt);
    const float t2 = key2f(kth) - two_e;
    RDX_RSTAMP(3);
    // Verification of the scan's threshold T (score units). The hits are exactly the allowed rows with coarse >= T. The k best of
    // them have exact >= c_k - E, so the exact k-th best of the corpus is >= c_k - E and every true top-k row has coarse >=
    // c_k - 2E: all of those were emitted iff c_k - 2E >= T (and k hits exist at all). A threshold taken from k sampled rows
    // (k_tau with rank k) passes by construction; a SPECULATIVE one (rank < k: an estimate of where the corpus' k-th score
    // lies) passes unless the estimate was too high — then the query goes to the fallback passes, which use
    // the provable threshold. T = -inf (fewer than k sets sampled): every allowed row was emitted, nothing to verify.
    const float tq = tau[q] * inv_scale2;
    if (tq > -INFINITY && ((int64_t)m < (int64_t)k || t2 < tq)) {
        if (threadIdx.x == 0) {
            atomicAdd(&ctr->spec_fail, 1);
            exact_list[atomicAdd(&ctr->n_exact, 1)] = q;
        }
        return;
    }
    if (threadIdx.x == 0) n_p = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint2 e = list[i];
        if (__uint_as_float(e.x) >= t2) {
            const int pos = atomicAdd(&n_p, 1);
            if (pos < REFINE_PMAX) s_r[pos] = (int64_t)e.y;
        }
    }
    __syncthreads();
    const int p = n_p;
    if (threadIdx.x == 0) atomicAdd(&ctr->rescored, (unsigned long long)p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* q4 = reinterpret_cast<const float4*>(qhat + (int64_t)q * dim);
    const int n4 = dim >> 2, nw = (int)(blockDim.x >> 6);
    // exact re-score of `cnt` candidates, one wave per candidate row: row_at(i) = its row, put(i, s) takes its exact score
    auto rescore = [&](int cnt, auto row_at, auto put) __attribute__((always_inline)) {
        if (n4 <= 256) {
            // dim <= 1024: the query sits in registers as doubles (converted once per block, not once per candidate), a row's four
            // 1 KiB pieces are requested together and the NEXT candidate's pieces before this one is summed. Same products, same
            // order per lane, same butterfly as exact_score(): the same bits. (c3: k = 100, ~140 candidates per query = 9 dependent
            // round trips to random HBM rows per wave before.)
            int gi[4];
            bool gv[4];
            float4 qf[4];                                   // (kept as floats, widened at use: 16 registers instead of 32)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gv[u] = lane + 64 * u < n4;
                gi[u] = gv[u] ? lane + 64 * u : n4 - 1;
                const float4 qq = q4[gi[u]];
                qf[u] = gv[u] ? qq : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            auto load_row = [&](int i, float4 (&c)[4]) __attribute__((always_inline)) {
                const MasterRow row4 = master_row(master, row_at(i), dim);
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = row4[gi[u]];
            };
            float4 cur[4], nxt[4];
            if (wave < cnt) load_row(wave, cur);
            for (int i = wave; i < cnt; i += nw) {
                load_row(i + nw < cnt ? i + nw : i, nxt);   // (the last one re-reads its own row: no branch around loads)
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc += (double)qf[u].x * (double)cur[u].x;
                    acc += (double)qf[u].y * (double)cur[u].y;
                    acc += (double)qf[u].z * (double)cur[u].z;
                    acc += (double)qf[u].w * (double)cur[u].w;
                }
                const float sx = (float)wave_sum(acc);
                if (lane == 0) put(i, sx);
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            }
            // (a third row in flight per wave — the loop is 9 dependent ~2 us round trips at c3 — needs 16 more registers than the
            //  1024-thread block has: 17 spilled. Measured instead: profiles/r04/refine_stamps.txt)
        } else {
            for (int i = wave; i < cnt; i += nw) {
                const float sx = exact_score(master_row(master, row_at(i), dim), q4, n4, lane);
                if (lane == 0) put(i, sx);
            }
        }
    };
    if (p > REFINE_PMAX) {
        // More rows inside the 2E band than the ranking arrays hold: near-duplicate rows stored together (a document's chunks: their
        // scores lie closer together than the coarse pass can tell apart, so the band holds hundreds to thousands of them). Until
        // round 4 such a query paid the exact full scan of the WHOLE corpus (7 ms per 4 queries at 10 M rows: an embedding-like corpus
        // ran at 2 % of the N(0,1) corpus' speed). The band is small next to the corpus: its members are compacted to the front of the
        // list (every thread reads its <= 7 entries, then — behind a barrier — writes the members back), re-scored exactly in place
        // (exact score over coarse score), the k-th largest exact score is found by the radix select, and only what reaches it is ranked.
        __shared__ int wtot[16];
        const int c = (int)((m + blockDim.x - 1) / blockDim.x);      // <= REFINE_LIST / 1024 = 7
        uint2 mine[7];
        bool keep[7];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t i = threadIdx.x * c + j;
            keep[j] = false;
            if (j < c && i < m) {
                mine[j] = list[i];
                keep[j] = __uint_as_float(mine[j].x) >= t2;
            }
            cnt += keep[j] ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();                                              // (also: every thread has read its chunk)
        int pos = incl - cnt;
        for (int w = 0; w < wave; ++w) pos += wtot[w];
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (keep[j]) list[pos++] = mine[j];
        __syncthreads();
        rescore(p, [&](int i) { return (int64_t)list[i].y; }, [&](int i, float sx) { list[i].x = __float_as_uint(sx); });
        __syncthreads();
        const int64_t kk2 = k < p ? k : p;
        int64_t n_gt2;
        const uint32_t kth2 = block_kth_largest([&](int64_t i) { return f2key(__uint_as_float(list[i].x)); }, p, kk2, hist, bc, &n_gt2);
        if (threadIdx.x == 0) n_p = 0;
        __syncthreads();
        // everything above the k-th key, and EVERY entry equal to it (identical rows tie: the ranking below orders them by row id)
        for (int i = threadIdx.x; i < p; i += blockDim.x) {
            const uint2 e = list[i];
            if (f2key(__uint_as_float(e.x)) >= kth2) {
                const int at = atomicAdd(&n_p, 1);
                if (at < REFINE_PMAX) {
                    s_s[at] = __uint_as_float(e.x);
                    s_r[at] = (int64_t)e.y;
                }
            }
        }
        __syncthreads();
        const int p2 = n_p;
        if (p2 > REFINE_PMAX) {                              // > 1024 - k rows IDENTICAL to the k-th: only the exact scan orders those
            if (threadIdx.x == 0) exact_list[atomicAdd(&ctr->n_exact, 1)] = q;
            return;
        }
        for (int i = threadIdx.x; i < p2; i += blockDim.x) s_r[i] = row_map ? row_map[s_r[i]] : s_r[i] + row_base;
        __syncthreads();
        rank_and_write(s_s, s_r, p2, k, o_s, o_r, out_count + q);
        return;
    }
    RDX_RSTAMP(4);
    rescore(p, [&](int i) { return s_r[i]; }, [&](int i, float sx) { s_s[i] = sx; });
    __syncthreads();
    RDX_RSTAMP(5);
    // local -> returned row id: + row_base, or through the shard's (strictly increasing) row id map
    for (int i = threadIdx.x; i < p; i += blockDim.x) s_r[i] = row_map ? row_map[s_r[i]] : s_r[i] + row_base;
    __syncthreads();
    rank_and_write(s_s, s_r, p, k, o_s, o_r, out_count + q);
    RDX_RSTAMP(6);
}

__global__ __launch_bounds__(1024) void k_refine(const uint2* __restrict__ cand, const uint32_t* __restrict__ cntw,
                                                int n_streams, uint32_t capw, uint32_t list_cap, int k, float two_e,
                                                const float* __restrict__ qhat, MasterView master, int dim,
                                                int64_t row_base, const int64_t* __restrict__ row_map,
                                                float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                int32_t* __restrict__ out_count, int32_t* __restrict__ exact_list,
                                                RefineCounters* __restrict__ ctr, const float* __restrict__ tau, float inv_scale2,
                                                const FinishArgs fin) {
    refine_query(cand, cntw, n_streams, capw, list_cap, k, two_e, qhat, master, dim, row_base, row_map, out_score, out_row, out_count,
                 exact_list, ctr, tau, inv_scale2);
    finish_if_last(fin);
}

// K5b as a kernel (the per-query work is select_dense_query, k_rows.hpp): the last launch of an exact-path search ends it
__global__ __launch_bounds__(1024) void k_select_dense(const float* __restrict__ scores, int64_t rows,
                                                       const int32_t* __restrict__ q_list, int k, int64_t row_base,
                                                       const int64_t* __restrict__ row_map,
                                                       float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                       int32_t* __restrict__ out_count, unsigned long long* __restrict__ t_last,
                                                       const FinishArgs fin) {
    select_dense_query(scores, rows, q_list, k, row_base, row_map, out_score, out_row, out_count, t_last);
    finish_if_last(fin);
}

// Second chance for queries whose candidate segments overflowed: their raw vectors are gathered into a small batch
// (one wave per query) ...
__global__ __launch_bounds__(256) void k_gather_queries(const float* __restrict__ queries, const int32_t* __restrict__ list, int m,
                                                        int dim, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= m) return;
    const float4* src = reinterpret_cast<const float4*>(queries + (int64_t)list[i] * dim);
    float4* dst = reinterpret_cast<float4*>(out + (int64_t)i * dim);
    for (int g = lane; g < (dim >> 2); g += 64) dst[g] = src[g];
}
// ... and the batch's answers are put back where those queries belong
__global__ __launch_bounds__(256) void k_scatter_topk(const float* __restrict__ s, const int64_t* __restrict__ r,
                                                      const int32_t* __restrict__ c, const int32_t* __restrict__ list, int m, int k,
                                                      float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                      int32_t* __restrict__ out_count) {
    const int i = blockIdx.x;
    if (i >= m) return;
    const int64_t q = list[i];
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        out_score[q * k + j] = s[(int64_t)i * k + j];
        out_row[q * k + j] = r[(int64_t)i * k + j];
    }
    if (threadIdx.x == 0) out_count[q] = c[i];
}

// C1. merge n_parts partial top-k lists per query. One block per query; n_parts*k <= MERGE_MAX.
// part p of each array starts stride_* ELEMENTS after part p-1 (contiguous [n_parts][nq][k] arrays, or the
// packed all-gather receive buffer).
constexpr int MERGE_MAX = 4096;
__global__ __launch_bounds__(256) void k_merge(const float* __restrict__ part_score, const int64_t* __restrict__ part_row,
                                               const int32_t* __restrict__ part_count, int64_t stride_s, int64_t stride_r,
                                               int64_t stride_c, int n_parts, int64_t nq, int k,
                                               float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                               int32_t* __restrict__ out_count, const int32_t* __restrict__ part_flags = nullptr,
                                               int64_t stride_f = 0, unsigned long long* __restrict__ sig_word = nullptr,
                                               unsigned long long sig_seq = 0) {
    // rdx_signal: the first block tells the host, before it merges anything, whether some rank's partial carried the
    // "incomplete" flag (one 64-bit word, sequence number and value together: nothing to order)
    if (sig_word && blockIdx.x == 0 && threadIdx.x == 0) {
        int any = 0;
        for (int p = 0; p < n_parts; ++p) any |= part_flags[(int64_t)p * stride_f] != 0;
        __hip_atomic_store(sig_word, (sig_seq << 1) | (unsigned long long)any, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ float s_s[MERGE_MAX];
    __shared__ int64_t s_r[MERGE_MAX];
    __shared__ int base[65];
    const int64_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int p = 0; p < n_parts; ++p) {
            base[p] = acc;
            int c = part_count[(int64_t)p * stride_c + q];
            acc += c < 0 ? 0 : (c > k ? k : c);
        }
        base[n_parts] = acc;
    }
    __syncthreads();
    // one thread per (part, slot): every load of the gather in flight at once (part after part, k threads at a time, it was
    // n_parts dependent round trips)
    for (int t = threadIdx.x; t < n_parts * k; t += blockDim.x) {
        const int p = t / k, i = t - p * k;
        if (i < base[p + 1] - base[p]) {
            s_s[base[p] + i] = part_score[(int64_t)p * stride_s + q * k + i];
            s_r[base[p] + i] = part_row[(int64_t)p * stride_r + q * k + i];
        }
    }
    __syncthreads();
    rank_and_write(s_s, s_r, base[n_parts], k, out_score + q * k, out_row + q * k, out_count + q);
}

// C1b. merge of TWO partial lists per query when n_parts * k exceeds what k_merge ranks in LDS (k > 2048 with two parts, ...):
// both lists are sorted (score desc, row asc) and no row occurs twice (row ids are global), so an element's place in the merged
// list is its index in its own list plus the number of elements of the other list that come before it — one binary search per
// element, no LDS. rdx_merge_topk folds the parts through this kernel one after the other. out must not alias a or b.
__global__ __launch_bounds__(256) void k_merge_pair(const float* __restrict__ a_s, const int64_t* __restrict__ a_r,
                                                    const int32_t* __restrict__ a_c, const float* __restrict__ b_s,
                                                    const int64_t* __restrict__ b_r, const int32_t* __restrict__ b_c, int k,
                                                    float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                    int32_t* __restrict__ out_count) {
    const int64_t q = blockIdx.x;
    const int na = min(max(a_c[q], 0), k), nb = min(max(b_c[q], 0), k);
    const float* as = a_s + q * k;
    const float* bs = b_s + q * k;
    const int64_t* ar = a_r + q * k;
    const int64_t* br = b_r + q * k;
    float* os = out_score + q * k;
    int64_t* orow = out_row + q * k;
    for (int t = threadIdx.x; t < na + nb; t += blockDim.x) {
        const bool from_a = t < na;
        const int i = from_a ? t : t - na;
        const float s = from_a ? as[i] : bs[i];
        const int64_t r = from_a ? ar[i] : br[i];
        const float* xs = from_a ? bs : as;
        const int64_t* xr = from_a ? br : ar;
        int lo = 0, hi = from_a ? nb : na;   // first element of the other list that does NOT come before (s, r)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float ms = xs[mid];
            const bool before = ms > s || (ms == s && xr[mid] < r);
            if (before) lo = mid + 1;
            else hi = mid;
        }
        const int pos = i + lo;
        if (pos < k) {
            os[pos] = s;
            orow[pos] = r;
        }
    }
    const int c = na + nb < k ? na + nb : k;
    for (int i = c + threadIdx.x; i < k; i += blockDim.x) {
        os[i] = -INFINITY;
        orow[i] = -1;
    }
    if (threadIdx.x == 0) out_count[q] = c;
}

}  // namespace rdx
