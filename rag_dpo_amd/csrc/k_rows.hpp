// k_rows.hpp — row-wise kernels: K1 L2-normalise (+ tiled fp16 scan copy), gather, exact scoring, dense select.
// One 64-lane wavefront owns one row: 16 B/lane coalesced loads, fp64 lane-order sums (oracle/rdx_oracle.c).
#pragma once
#include "rdx_common.hpp"

namespace rdx {

// The exact copy of the corpus rows ("master"): what the exact re-score K4, the exact scan K5 and get() read.
//   default : fp32 [cap][dim], the L2-normalised rows (4 B/element)
//   compact : the raw bf16 rows as delivered (2 B/element) + each row's divisor den = max(|x|_2, 1e-12) as a double. The
//             normalised element y_i = (float)((double)x_i / den) is RECOMPUTED wherever it is needed — the same two
//             operands and the same operation K1 uses, hence the same bits (oracle/rdx_oracle.c) — instead of being stored:
//             a bf16 corpus (BASELINE config 5) then costs 2 (raw) + 2 (fp16 scan copy) B/element instead of 4 + 2.
struct MasterView {
    float* f32;
    uint16_t* raw16;
    double* den;
};
struct MasterRow {
    const float4* p32;
    const ushort4* p16;
    double den;
    __device__ __forceinline__ float4 operator[](int g) const {
        if (p32) return p32[g];
        const ushort4 u = p16[g];
        float4 y;
        y.x = (float)((double)__uint_as_float((uint32_t)u.x << 16) / den);
        y.y = (float)((double)__uint_as_float((uint32_t)u.y << 16) / den);
        y.z = (float)((double)__uint_as_float((uint32_t)u.z << 16) / den);
        y.w = (float)((double)__uint_as_float((uint32_t)u.w << 16) / den);
        return y;
    }
};
__device__ __forceinline__ MasterRow master_row(const MasterView& mv, int64_t r, int dim) {
    MasterRow m;
    m.p32 = mv.f32 ? reinterpret_cast<const float4*>(mv.f32 + r * (int64_t)dim) : nullptr;
    m.p16 = mv.f32 ? nullptr : reinterpret_cast<const ushort4*>(mv.raw16 + r * (int64_t)dim);
    m.den = mv.f32 ? 1.0 : mv.den[r];
    return m;
}

// write the 4 consecutive elements k..k+3 of a normalised row into a scan copy (corpus: fragment order, queries: LDS images)
template <bool QUERY>
__device__ __forceinline__ void shadow_store4(_Float16* __restrict__ shadow, int64_t row, int k, int ksteps,
                                              float scale, float4 y) {
    half4 h;
    h[0] = (_Float16)(y.x * scale);   // RNE; y * 2^s is exact
    h[1] = (_Float16)(y.y * scale);
    h[2] = (_Float16)(y.z * scale);
    h[3] = (_Float16)(y.w * scale);
    *reinterpret_cast<half4*>(shadow + (QUERY ? query_off(row, k, ksteps) : corpus_off(row, k, ksteps))) = h;
}

// K1. out = in / max(|in|_2, 1e-12)  — SentenceTransformer.encode(normalize_embeddings=True),
// reference src/utils/embedding_provider.py:139-145; also what `collection.add` needs for the cosine
// space (reference src/processing/create_chromadb_index.py:100-106,374-379).
// Destination row of input row i is dst_rows ? dst_rows[i] : row0 + i.
template <bool QUERY>
__global__ __launch_bounds__(256) void k_normalize(const float* __restrict__ in, const uint16_t* __restrict__ in_bf16,
                                                   int64_t n, int dim, const int64_t* __restrict__ dst_rows,
                                                   int64_t row0, MasterView master,
                                                   _Float16* __restrict__ shadow, int ksteps, float scale,
                                                   int* __restrict__ bad, int64_t n_pad = 0, int verbatim = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if constexpr (QUERY) {
        // the query scan copy is rebuilt for every search: padding rows [n, n_pad) and padding columns [dim, 64*ksteps) are
        // written as zeros here (no separate memset launch)
        if (i >= n) {
            if (i < n_pad && shadow)
                for (int g = lane; g < ksteps * 16; g += 64) shadow_store4<true>(shadow, row0 + i, 4 * g, ksteps, scale, make_float4(0.f, 0.f, 0.f, 0.f));
            return;
        }
        if (shadow)
            for (int g = (dim >> 2) + lane; g < ksteps * 16; g += 64)
                shadow_store4<true>(shadow, row0 + i, 4 * g, ksteps, scale, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (i >= n) return;
    const int n4 = dim >> 2;
    auto load4 = [&](int g) -> float4 {
        if (in_bf16) {
            const ushort4 u = reinterpret_cast<const ushort4*>(in_bf16 + i * (int64_t)dim)[g];
            return make_float4(__uint_as_float((uint32_t)u.x << 16), __uint_as_float((uint32_t)u.y << 16),
                               __uint_as_float((uint32_t)u.z << 16), __uint_as_float((uint32_t)u.w << 16));
        }
        return reinterpret_cast<const float4*>(in + i * (int64_t)dim)[g];
    };
    // dim <= 1024: the row's (up to) four float4 groups per lane are requested together and kept in registers for the second
    // pass — one memory round trip instead of four dependent ones plus four re-reads (a query batch is a handful of rows: the
    // kernel is all latency). Same additions in the same order per lane, same butterfly: the same bits.
    const bool in_regs = n4 <= 256;
    float4 rv[4];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = load4(lane + 64 * u < n4 ? lane + 64 * u : n4 - 1);
    }
    double acc = 0.0;
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (lane + 64 * u < n4) {
                acc += (double)rv[u].x * (double)rv[u].x;
                acc += (double)rv[u].y * (double)rv[u].y;
                acc += (double)rv[u].z * (double)rv[u].z;
                acc += (double)rv[u].w * (double)rv[u].w;
            }
        }
    } else {
        for (int g = lane; g < n4; g += 64) {
            const float4 v = load4(g);
            acc += (double)v.x * (double)v.x;
            acc += (double)v.y * (double)v.y;
            acc += (double)v.z * (double)v.z;
            acc += (double)v.w * (double)v.w;
        }
    }
    const double n2 = wave_sum(acc);
    if (!(n2 < 1.0e300)) {   // NaN or Inf somewhere in the row: reject the whole call
        if (lane == 0) atomicOr(bad, 1);
        return;
    }
    double den = sqrt(n2);
    if (den < 1e-12) den = 1e-12;
    if (verbatim) den = 1.0;   // rows that ARE stored values (snapshot reload): x / 1.0 == x, kept bit for bit
    const int64_t dst = dst_rows ? dst_rows[i] : row0 + i;
    if (master.raw16 && lane == 0) master.den[dst] = den;   // compact master: the raw row + its divisor
    auto emit = [&](int g, const float4& v) __attribute__((always_inline)) {
        float4 y;
        y.x = (float)((double)v.x / den);
        y.y = (float)((double)v.y / den);
        y.z = (float)((double)v.z / den);
        y.w = (float)((double)v.w / den);
        if (master.f32) reinterpret_cast<float4*>(master.f32 + dst * (int64_t)dim)[g] = y;
        if (master.raw16) reinterpret_cast<ushort4*>(master.raw16 + dst * (int64_t)dim)[g] = reinterpret_cast<const ushort4*>(in_bf16 + i * (int64_t)dim)[g];
        if (shadow) shadow_store4<QUERY>(shadow, dst, 4 * g, ksteps, scale, y);
    };
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < n4) emit(lane + 64 * u, rv[u]);
    } else {
        for (int g = lane; g < n4; g += 64) emit(g, load4(g));
    }
}

// row_map[first + i] = base + first + i (identity part of a shard's local -> global row id map)
__global__ __launch_bounds__(256) void k_iota64(int64_t* __restrict__ out, int64_t first, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[first + i] = base + first + i;
}

// rebuild the scan copy of rows [row0, row0+n) from the (already normalised) master copy
__global__ __launch_bounds__(256) void k_reshadow(MasterView master, int64_t row0, int64_t n, int dim,
                                                  _Float16* __restrict__ shadow, int ksteps, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = row0 + i;
    const int n4 = dim >> 2;
    const MasterRow row = master_row(master, r, dim);
    for (int g = lane; g < n4; g += 64) shadow_store4<false>(shadow, r, 4 * g, ksteps, scale, row[g]);
}

// out[i] = master[rows[i]]  (collection.get(include=["embeddings"]), compaction)
__global__ __launch_bounds__(256) void k_gather_rows(MasterView master, const int64_t* __restrict__ rows,
                                                     int64_t n, int dim, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const MasterRow src = master_row(master, rows[i], dim);
    float4* dst = reinterpret_cast<float4*>(out + i * (int64_t)dim);
    for (int g = lane; g < (dim >> 2); g += 64) dst[g] = src[g];
}

// compaction of a compact master: raw rows and divisors move as they are
__global__ __launch_bounds__(256) void k_gather_raw(MasterView master, const int64_t* __restrict__ rows, int64_t n, int dim,
                                                    MasterView out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = rows[i];
    const ushort4* src = reinterpret_cast<const ushort4*>(master.raw16 + r * (int64_t)dim);
    ushort4* dst = reinterpret_cast<ushort4*>(out.raw16 + i * (int64_t)dim);
    for (int g = lane; g < (dim >> 2); g += 64) dst[g] = src[g];
    if (lane == 0) out.den[i] = master.den[r];
}

// exact score of one normalised query (float4 view, global or LDS) against one master row, all 64 lanes get it
template <class Q4>
__device__ __forceinline__ float exact_score(const MasterRow& row4, Q4 q4, int n4, int lane) {
    double acc = 0.0;
    for (int g = lane; g < n4; g += 64) {
        const float4 c = row4[g];
        const float4 q = q4[g];
        acc += (double)q.x * (double)c.x;
        acc += (double)q.y * (double)c.y;
        acc += (double)q.z * (double)c.z;
        acc += (double)q.w * (double)c.w;
    }
    return (float)wave_sum(acc);
}

// K5a. Exact scores of up to QX queries against EVERY row (small problems, huge k, overflow fallback):
// out[j][r] = score(q_list[j], r), or -inf when the row fails the `where` pre-filter.
constexpr int QX = 4;
__global__ __launch_bounds__(256) void k_exact_scores(MasterView master, int64_t rows, int dim,
                                                      const float* __restrict__ qhat, const int32_t* __restrict__ q_list,
                                                      int nq, const uint32_t* __restrict__ allow,
                                                      float* __restrict__ out, unsigned long long* __restrict__ t_first_inv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (t_first_inv && threadIdx.x == 0) atomicMax(t_first_inv, ~wall_clock64());   // profile = 3: earliest block start
    float4* q4 = reinterpret_cast<float4*>(smem);   // [nq][dim/4]
    const int n4 = dim >> 2;
    for (int j = 0; j < nq; ++j) {
        const float4* src = reinterpret_cast<const float4*>(qhat + (int64_t)q_list[j] * dim);
        for (int g = threadIdx.x; g < n4; g += blockDim.x) q4[j * n4 + g] = src[g];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    // A wave owns rows wave0, wave0 + nwaves, ...; the loads of a row's next four float4 groups are all issued before the
    // first of them is used (four independent 1 KiB requests in flight per wave instead of one: the kernel is latency-bound
    // at the reference's 16,919 rows). Per lane the groups are still added in increasing g: the oracle's lane order.
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const bool ok = !allow || ((allow[r >> 5] >> (r & 31)) & 1u);
        const MasterRow row4 = master_row(master, r, dim);
        double acc[QX];
#pragma unroll
        for (int j = 0; j < QX; ++j) acc[j] = 0.0;
        if (ok) {
            for (int g0 = lane; g0 < n4; g0 += 256) {
                float4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = g0 + 64 * u < n4 ? row4[g0 + 64 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (g0 + 64 * u < n4) {
#pragma unroll
                        for (int j = 0; j < QX; ++j) {
                            if (j < nq) {
                                const float4 q = q4[j * n4 + g0 + 64 * u];
                                acc[j] += (double)q.x * (double)c[u].x;
                                acc[j] += (double)q.y * (double)c[u].y;
                                acc[j] += (double)q.z * (double)c[u].z;
                                acc[j] += (double)q.w * (double)c[u].w;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < QX; ++j) {
            if (j < nq) {
                const float s = (float)wave_sum(acc[j]);
                if (lane == 0) out[(int64_t)j * rows + r] = ok ? s : -INFINITY;
            }
        }
    }
}

// K5a for dim <= 1024 (U = float4 groups per lane, 1..4): the same scores, bit for bit, with three changes that matter at the
// reference's own shape (16,919 rows, 4 queries, where this kernel IS the search). (1) The queries live in REGISTERS as
// doubles (64 per lane at d = 1024) instead of being re-read from LDS and re-converted for every row: the grid is small enough
// (2 blocks per CU) that a wave keeps them over ~8 rows. (2) The next row's loads are issued before the current row is
// summed (two rows in flight per wave). (3) The four per-query butterflies run as ONE transposed butterfly: after the m = 32
// stage a lane keeps two of the four partial sums, after m = 16 one, so 14 cross-lane dword moves and 7 additions replace 48
// and 24 — every addition pairs the same two values as wave_sum's butterfly (lane order, oracle/rdx_oracle.c), so the bits
// are the same. The sum of query j ends on lanes with (lane >> 4) == j.
template <int U>
__global__ __launch_bounds__(256, 2) void k_exact_scores_reg(MasterView master, int64_t rows, int dim,
                                                             const float* __restrict__ qhat, const int32_t* __restrict__ q_list,
                                                             int nq, const uint32_t* __restrict__ allow, float* __restrict__ out,
                                                             unsigned long long* __restrict__ t_first_inv) {
    if (t_first_inv && threadIdx.x == 0) atomicMax(t_first_inv, ~wall_clock64());   // profile = 3: earliest block start
    const int n4 = dim >> 2;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    if (wave0 >= rows) return;
    int gi[U];        // this lane's float4 groups (clamped: a group past the row's end re-reads the last one and counts as zeros)
    bool gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        gv[u] = lane + 64 * u < n4;
        gi[u] = gv[u] ? lane + 64 * u : n4 - 1;
    }
    auto load_row = [&](int64_t r, float4 (&c)[U]) __attribute__((always_inline)) {
        const MasterRow row4 = master_row(master, r, dim);
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = row4[gi[u]];
    };
    float4 cur[U], nxt[U];
    load_row(wave0, cur);
    double qd[QX][U][4];
#pragma unroll
    for (int j = 0; j < QX; ++j) {
        const float4* src = reinterpret_cast<const float4*>(qhat + (int64_t)q_list[j < nq ? j : 0] * dim);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 q = src[gi[u]];
            const bool on = j < nq && gv[u];
            qd[j][u][0] = on ? (double)q.x : 0.0;
            qd[j][u][1] = on ? (double)q.y : 0.0;
            qd[j][u][2] = on ? (double)q.z : 0.0;
            qd[j][u][3] = on ? (double)q.w : 0.0;
        }
    }
    const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const int64_t rn = r + nwaves < rows ? r + nwaves : r;   // (the last iteration re-reads its own row: no branch around loads)
        load_row(rn, nxt);
        const bool ok = !allow || ((allow[r >> 5] >> (r & 31)) & 1u);
        double acc[QX];
#pragma unroll
        for (int j = 0; j < QX; ++j) acc[j] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double cx = (double)cur[u].x, cy = (double)cur[u].y, cz = (double)cur[u].z, cw = (double)cur[u].w;
#pragma unroll
            for (int j = 0; j < QX; ++j) {
                acc[j] += qd[j][u][0] * cx;   // (products of two floats are exact in double: fused or not, the same bits)
                acc[j] += qd[j][u][1] * cy;
                acc[j] += qd[j][u][2] * cz;
                acc[j] += qd[j][u][3] * cw;
            }
        }
        // transposed butterfly: m = 32 (keep queries {0,1} on lanes < 32, {2,3} on the others), m = 16, then m = 8..1 on the one left
        const double s0 = hi32 ? acc[0] : acc[2], s1 = hi32 ? acc[1] : acc[3];   // what the partner keeps
        double k0 = hi32 ? acc[2] : acc[0], k1 = hi32 ? acc[3] : acc[1];
        k0 += __shfl_xor(s0, 32, 64);
        k1 += __shfl_xor(s1, 32, 64);
        const double s2 = hi16 ? k0 : k1;
        double v = hi16 ? k1 : k0;
        v += __shfl_xor(s2, 16, 64);
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        const int j = lane >> 4;
        if ((lane & 15) == 0 && j < nq) out[(int64_t)j * rows + r] = ok ? (float)v : -INFINITY;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

// rank entries (score desc, row asc) held in LDS and write the best k; all threads of the block call it.
// Wave-parallel: a wave takes entry i, its lanes take the entries j it is compared with (64 at a time), the rank is the
// population count of the ballots — p/64 LDS reads per entry instead of p dependent ones per thread (p = 50: 5.2 -> ~1 us in
// k_select_dense; the same routine ends k_refine and k_merge).
__device__ __forceinline__ void rank_and_write(const float* __restrict__ s, const int64_t* __restrict__ r, int p, int k,
                                               float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                               int32_t* __restrict__ out_count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = wave; i < p; i += nw) {
        const float si = s[i];
        const int64_t ri = r[i];
        int rank = 0;
        for (int j0 = 0; j0 < p; j0 += 64) {
            const int j = j0 + lane;
            bool before = false;
            if (j < p) {
                const float sj = s[j];
                before = (sj > si) || (sj == si && r[j] < ri);
            }
            rank += __popcll(__ballot(before));
        }
        if (lane == 0 && rank < k) {
            out_score[rank] = si;
            out_row[rank] = ri;
        }
    }
    const int c = p < k ? p : k;
    for (int i = c + threadIdx.x; i < k; i += blockDim.x) {
        out_score[i] = -INFINITY;
        out_row[i] = -1;
    }
    if (threadIdx.x == 0) *out_count = c;
}

// K5b. top-k of a dense score row (one block per listed query): radix select of the k-th largest score, then
// everything above it plus the LOWEST-row entries equal to it (ties -> ascending row id), then a rank sort.
// -inf marks rows excluded by the `where` pre-filter; they are never returned.
constexpr int SELECT_MAX_K = 4096;
#ifdef RDX_SELECT_STAMPS   // developer build (tools/select_stamps.py): where k_select_dense spends its time
__device__ unsigned long long g_select_stamps[16];
#define RDX_STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_select_stamps[i] = wall_clock64(); } while (0)
#else
#define RDX_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void select_dense_query(const float* __restrict__ scores, int64_t rows,
                                                   const int32_t* __restrict__ q_list, int k, int64_t row_base,
                                                   const int64_t* __restrict__ row_map,
                                                   float* __restrict__ out_score, int64_t* __restrict__ out_row,
                                                   int32_t* __restrict__ out_count, unsigned long long* __restrict__ t_last) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[HIST_WORDS];
    __shared__ uint32_t bc[4];
    __shared__ float s_s[SELECT_MAX_K];
    __shared__ int64_t s_r[SELECT_MAX_K];
    __shared__ int n_sel, n_eq_taken, wave_tot[16];
    const int j = blockIdx.x;
    const int q = q_list[j];
    const float* sc = scores + (int64_t)j * rows;
    float* o_s = out_score + (int64_t)q * k;
    int64_t* o_r = out_row + (int64_t)q * k;
    if (rows == 0 || k == 0) {
        rank_and_write(s_s, s_r, 0, k, o_s, o_r, out_count + q);
        return;
    }
    RDX_STAMP(0);
    const int64_t kk = k < rows ? k : rows;
    int64_t n_gt;
    // Up to 32 Ki rows the block keeps the whole score row in REGISTERS (32 keys per thread, loaded once); longer rows are
    // re-read from global memory in every pass.
    constexpr int RN = 32;
    const bool in_regs = rows <= (int64_t)RN * 1024 && blockDim.x == 1024;
    uint32_t kreg[RN];
    if (in_regs) {
        // unconditional loads (clamped index), all in flight at once: written as "value or padding" the compiler guards every
        // load with its own branch and waits for each (17 L2 round trips at the reference's 16,919 rows: 5 us)
        float vreg[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int64_t i = (int64_t)j * 1024 + threadIdx.x;
            vreg[j] = sc[i < rows ? i : rows - 1];
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) asm volatile("" : "+v"(vreg[j]));   // the loads stay where they are
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int64_t i = (int64_t)j * 1024 + threadIdx.x;
            kreg[j] = i < rows ? f2key(vreg[j]) : 0u;   // padding key 0 sorts below every real score (finite or -inf)
        }
    }
    RDX_STAMP(1);
    const uint32_t kth = in_regs ? block_kth_largest_scan(
                                       [&](auto f) {
#pragma unroll
                                           for (int j = 0; j < RN; ++j)
                                               if ((int64_t)j * 1024 + threadIdx.x < rows) f(kreg[j]);
                                       },
                                       kk, hist, bc, &n_gt)
                                 : block_kth_largest([&](int64_t i) { return f2key(sc[i]); }, rows, kk, hist, bc, &n_gt);
    RDX_STAMP(2);
    const int need_eq = (int)(kk - n_gt);   // >= 1
    constexpr int EQ_CAP = 1024;
    __shared__ int64_t eq_idx[EQ_CAP];
    __shared__ int n_eq;
    if (threadIdx.x == 0) {
        n_sel = 0;
        n_eq_taken = 0;
        n_eq = 0;
    }
    __syncthreads();
    // ONE more pass: entries strictly above the k-th key go to the list in any order; entries equal to it (normally one)
    // are collected on the side
    auto collect = [&](int64_t i, uint32_t key) {
        if (key > kth) {
            const int pos = atomicAdd(&n_sel, 1);
            s_s[pos] = key2f(key);
            s_r[pos] = row_map ? row_map[i] : row_base + i;
        } else if (key == kth) {
            const int e = atomicAdd(&n_eq, 1);
            if (e < EQ_CAP) eq_idx[e] = i;
        }
    };
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int64_t i = (int64_t)j * 1024 + threadIdx.x;
            if (i < rows) collect(i, kreg[j]);
        }
    } else {
        for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) collect(i, f2key(sc[i]));
    }
    __syncthreads();
    RDX_STAMP(3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (n_eq <= EQ_CAP) {
        // of the equal entries the need_eq with the LOWEST rows belong to the answer (ties -> ascending row id)
        const int ne = n_eq;
        for (int e = threadIdx.x; e < ne; e += blockDim.x) {
            const int64_t i = eq_idx[e];
            int rank = 0;
            for (int f = 0; f < ne; ++f) rank += eq_idx[f] < i;
            if (rank < need_eq) {
                s_s[(int)n_gt + rank] = sc[i];
                s_r[(int)n_gt + rank] = row_map ? row_map[i] : row_base + i;
            }
        }
    } else {
        // thousands of identical scores: walk the rows in order until need_eq equal entries are taken
        for (int64_t base = 0; base < rows; base += blockDim.x) {
            if (n_eq_taken >= need_eq) break;   // uniform: read after the barrier below
            const int64_t i = base + threadIdx.x;
            const bool eq = i < rows && f2key(sc[i]) == kth;
            const unsigned long long m = __ballot(eq);
            if (lane == 0) wave_tot[wave] = __popcll(m);
            __syncthreads();
            int before = n_eq_taken;
            for (int w = 0; w < wave; ++w) before += wave_tot[w];
            const int my = before + __popcll(m & ((1ull << lane) - 1ull));
            if (eq && my < need_eq) {
                s_s[(int)n_gt + my] = sc[i];
                s_r[(int)n_gt + my] = row_map ? row_map[i] : row_base + i;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int t = 0;
                for (int w = 0; w < nw; ++w) t += wave_tot[w];
                n_eq_taken += t;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    RDX_STAMP(4);
    // drop -inf (filtered) entries: they sort last, so count the finite prefix after ranking
    int p = (int)kk;
    __shared__ int n_fin;
    if (threadIdx.x == 0) n_fin = 0;
    __syncthreads();
    int loc = 0;
    for (int i = threadIdx.x; i < p; i += blockDim.x) loc += (s_s[i] > -INFINITY);
    if (loc) atomicAdd(&n_fin, loc);
    __syncthreads();
    const int valid = n_fin;
    RDX_STAMP(5);
    rank_and_write(s_s, s_r, p, valid < k ? valid : k, o_s, o_r, out_count + q);
    if (t_last) {   // profile = 3: latest block end
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(t_last, wall_clock64());
    }
    RDX_STAMP(6);
    // rank_and_write filled [valid, k) only up to its own k argument; pad the rest
    for (int i = valid + threadIdx.x; i < k; i += blockDim.x) {
        o_s[i] = -INFINITY;
        o_r[i] = -1;
    }
}

}  // namespace rdx
