// k_scan.hpp — K2+K3: the corpus scan. fp16 MFMA GEMM  S[row][query] = <corpus row, query>  over the tiled
// scan copy, with the top-k work fused into the epilogue so the B x N score matrix never exists in HBM.
//
// Roofline: reads rows*dim_pad*2 B of corpus exactly once per launch (HBM) and does 2*rows*nq*dim_pad flop
// (MFMA). HBM-bound while nq per sweep is small, MFMA-bound from nq ~ 300 up.
//
// Work split. A workgroup (8 waves) owns a (256-row corpus tile) x (BN-query tile) block per step and walks
// a *stream* of corpus tiles; the nqt query tiles of one stream run on WGs with equal blockIdx % 8, i.e. on
// one XCD, so the corpus tile is fetched from HBM once and re-read from that XCD's L2 (speed only — nothing
// depends on the placement). Waves are laid out WM x WN with WN*64 = BN: a wave owns (256/WM) rows x 64
// queries = MR x 2 blocks of 32x32, MFMA v_mfma_f32_32x32x16_f16 with the corpus as the A operand and the
// queries as the B operand: D[row][query] has the QUERY on the lane (col = lane & 31) and 16 corpus rows in
// the 16 accumulator registers, so the per-query threshold lives in one register per lane and the epilogue
// is compare-only.
//
// Epilogues.
//   EPI_SETMAX (threshold bootstrap, run on every sample_div-th tile): each accumulator register position
//     keeps a running max over all tiles of the stream -> 32 disjoint row sets per (stream, wave row) and
//     query. k_tau takes the k-th largest set max: k DISTINCT rows score at least that, which makes
//     tau = that - 2E a lower bound for every true top-k row's coarse score (E = |coarse - exact| bound).
//   EPI_EMIT (main pass): every (row, query) whose coarse score >= tau[query] is appended to the query's
//     candidate list (global atomic slot counter; a few hundred hits per query over the whole corpus).
#pragma once
#include "rdx_common.hpp"

namespace rdx {

constexpr int EPI_SETMAX = 0;
constexpr int EPI_EMIT = 1;

struct ScanParams {
    const _Float16* shadow;    // tiled corpus scan copy
    const _Float16* qshadow;   // tiled query scan copy (same layout, row = query)
    int ksteps;                // dim_pad / 64
    int64_t rows;              // valid corpus rows
    int64_t n_tiles;           // ceil(rows / 256)
    int tile_stride;           // 1 (main) or sample_div (bootstrap): tiles 0, stride, 2*stride, ...
    int nqt;                   // query tiles of BN queries
    int nq_pad;                // queries padded to 256
    const uint32_t* allow;     // NULL or row bitmap
    // EPI_SETMAX
    float* setmax;             // [nq_pad][n_sets]
    int n_sets;
    // EPI_EMIT
    const float* tau;          // [nq_pad] in accumulator units (score * 4^scale_log2)
    uint32_t* cnt;             // [nq_pad]
    uint2* cand;               // [nq_pad][cap] (score bits, row)
    uint32_t cap;
    float inv_scale2;          // accumulator -> score
};

// row inside a 32x32 MFMA block held by accumulator register r of a lane in half h (= lane >> 5)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int BN, int EPI, bool HAS_MASK>
__global__ __launch_bounds__(512) void k_scan(const ScanParams p) {
    constexpr int WN = BN / 64;          // waves along queries
    constexpr int WM = 8 / WN;           // waves along corpus rows
    constexpr int MW = TILE_ROWS / WM;   // rows per wave
    constexpr int MR = MW / 32;          // 32-row blocks per wave
    constexpr int A_BYTES = KSTEP_BYTES;
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- which stream / query tile am I (XCD-aware: blocks with equal blockIdx % 8 share an L2) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int wpx = gridDim.x >> 3;
    const int G = wpx / p.nqt;           // streams per XCD
    if (slot >= G * p.nqt) return;
    const int qt = slot % p.nqt;
    const int stream = xcd * G + slot / p.nqt;
    const int n_streams = 8 * G;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int n_sched = (int)((p.n_tiles + p.tile_stride - 1) / p.tile_stride);   // tiles in this launch
    const int my_tiles = stream < n_sched ? (n_sched - stream + n_streams - 1) / n_streams : 0;
    const int KS = p.ksteps;
    const int total = my_tiles * KS;   // k-steps of this workgroup (host keeps tiles*ksteps < 2^31)

    const char* qbase = reinterpret_cast<const char*>(p.qshadow) +
                        ((int64_t)(qt * BN / 256) * KS) * KSTEP_BYTES + (int64_t)((qt * BN) % 256) * 128;

    // stage (it, ks) -> LDS buffer `buf`: 32 KiB corpus image + BN*128 B query image, both contiguous in HBM
    auto issue_stage = [&](int it, int ks, int buf) {
        const int64_t tile = (int64_t)(stream + it * n_streams) * p.tile_stride;
        char* dst = smem + buf * STAGE_BYTES;
        const char* asrc = reinterpret_cast<const char*>(p.shadow) + (tile * KS + ks) * (int64_t)KSTEP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
        }
        const char* bsrc = qbase + (int64_t)ks * KSTEP_BYTES;
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {
            const int c = wave * (BN / 64) + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + A_BYTES + c * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MR][2];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // per-lane epilogue state
    float runmax[2][16];
    float tau_l[2];
    const int qcol0 = qt * BN + wn * 64 + l31;   // query of n-block 0; n-block 1 is +32
    if constexpr (EPI == EPI_SETMAX) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) runmax[n][r] = -INFINITY;
    } else {
        tau_l[0] = p.tau[qcol0];
        tau_l[1] = p.tau[qcol0 + 32];
    }

    // LDS read offsets of this lane's fragments (row-dependent part; k sub-step adds the slot XOR)
    int a_off[MR], b_off[2], a_sw[MR], b_sw[2];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int row = wm * MW + m * 32 + l31;
        a_off[m] = row * 128;
        a_sw[m] = (row >> 1) & 7;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = wn * 64 + n * 32 + l31;   // row inside this WG's query tile image
        const int grow = (qt * BN) % 256 + row;   // row inside the 256-row shadow block (swizzle uses it)
        b_off[n] = A_BYTES + row * 128;
        b_sw[n] = (grow >> 1) & 7;
    }

    if (total > 0) issue_stage(0, 0, 0);
    int it = 0, ks = 0;          // tile iteration / k-step being computed
    int nit = 0, nks = 1;        // the step after it (prefetched)
    if (nks == KS) { nks = 0; nit = 1; }
    for (int s = 0; s < total; ++s) {
        __syncthreads();   // stage s has landed for every wave (the fence drains vmcnt) and compute(s-1) is done
        if (s + 1 < total) issue_stage(nit, nks, (s + 1) & 1);
        const char* st = smem + (s & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int chunk = kk * 2 + half;
            half8 af[MR], bf[2];
#pragma unroll
            for (int m = 0; m < MR; ++m) af[m] = *reinterpret_cast<const half8*>(st + a_off[m] + ((chunk ^ a_sw[m]) << 4));
#pragma unroll
            for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const half8*>(st + b_off[n] + ((chunk ^ b_sw[n]) << 4));
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
        const bool last_k = ks == KS - 1;
        const int it_done = it;
        it = nit; ks = nks;
        if (++nks == KS) { nks = 0; ++nit; }
        if (last_k) {
            // ---------------- tile epilogue ----------------
            const int64_t tile = (int64_t)(stream + it_done * n_streams) * p.tile_stride;
            const int64_t row_w = tile * TILE_ROWS + wm * MW;   // first row of this wave
            const bool ragged = (tile + 1) * TILE_ROWS > p.rows; // tile holds padding rows
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int64_t row_b = row_w + m * 32;            // first row of the 32x32 block
                uint32_t okbits = 0xffffffffu;                   // bit i: row row_b+i may be used
                if (ragged) {
                    const int64_t left = p.rows - row_b;
                    okbits = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
                }
                if constexpr (HAS_MASK) {
                    if (row_b < p.rows) okbits &= p.allow[row_b >> 5];
                }
                const bool filt = HAS_MASK || ragged;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if constexpr (EPI == EPI_SETMAX) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[m][n][r];
                            if (filt && !((okbits >> acc_row(r, half)) & 1u)) v = -INFINITY;
                            runmax[n][r] = fmaxf(runmax[n][r], v);
                        }
                    } else {
                        float mx = acc[m][n][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[m][n][r]);
                        if (__any(mx >= tau_l[n])) {
                            const int q = qcol0 + n * 32;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[m][n][r];
                                const int rib = acc_row(r, half);
                                if (v >= tau_l[n] && (!filt || ((okbits >> rib) & 1u))) {
                                    const uint32_t pos = atomicAdd(&p.cnt[q], 1u);
                                    if (pos < p.cap)
                                        p.cand[(int64_t)q * p.cap + pos] =
                                            make_uint2(__float_as_uint(v * p.inv_scale2), (uint32_t)(row_b + rib));
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
                }
            }
        }
    }

    if constexpr (EPI == EPI_SETMAX) {
        // set id = (stream*WM + wm)*32 + r*2 + half ; layout setmax[query][set]
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float* dst = p.setmax + (int64_t)(qcol0 + n * 32) * p.n_sets + (int64_t)(stream * WM + wm) * SETS_PER_WAVE + half;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 2] = runmax[n][r];
        }
    }
}

// K3a. tau[q] = (k-th largest of the query's set maxima) - 2E, in accumulator units; -inf if fewer than k
// non-empty sets exist (then every allowed row is emitted). One block per query (padding queries: +inf).
__global__ __launch_bounds__(256) void k_tau(const float* __restrict__ setmax, int n_sets, int k, float two_e_scaled,
                                             int nq, float* __restrict__ tau) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t bc[4];
    const int q = blockIdx.x;
    if (q >= nq) {   // padding query (zero vector): must never emit
        if (threadIdx.x == 0) tau[q] = INFINITY;
        return;
    }
    const float* sm = setmax + (int64_t)q * n_sets;
    if (k > n_sets) {
        if (threadIdx.x == 0) tau[q] = -INFINITY;
        return;
    }
    int64_t n_gt;
    const uint32_t key = block_kth_largest([&](int64_t i) { return f2key(sm[i]); }, n_sets, k, hist, bc, &n_gt);
    if (threadIdx.x == 0) {
        const float v = key2f(key);
        tau[q] = v > -INFINITY ? v - two_e_scaled : -INFINITY;
    }
}

}  // namespace rdx
