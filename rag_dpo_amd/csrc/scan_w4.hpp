// scan_w4.hpp — K2w: the main scan of large launches in the ONE-WAVE-PER-SIMD layout (developer experiment, option "wave_layout" = 1;
// the shipped kernel is k_scan in scan_kernel.hpp and stays the default).
//
// k_scan runs 8 waves per CU, each owning 32 corpus rows x 256 queries: every query fragment a wave reads from LDS feeds TWO MFMAs
// (the wave's two 16-row blocks). Round 3 priced "half the LDS reads per MFMA" by ablation at +3 % (7 % more clock) and
// found that the 4 x 2 layout that gets there by loading corpus fragments twice loses it again in the address path. The one layout
// that halves the reads without loading anything twice: 4 waves per CU, each owning 64 rows x 256 queries —
//   accumulators  4 row blocks x 16 query blocks x 4 = 256 registers (of the 512 a lone wave of a SIMD may use)
//   corpus        8 global_load_dwordx4 per wave and k-step (the wave's two 32-row blocks of the scan copy), two k-steps ahead
//   queries       one ds_read_b128 per FOUR MFMAs; the 32 KiB query image of a k-step arrives by LDS-DMA, 8 pieces of 1 KiB per wave
//   MFMA          128 per wave and k-step (2048 matrix-pipe cycles)
// What it gives up is the partner wave that covers every stall of the other one (DMA issue, the counted wait, the barrier, the emit
// check). To keep the lone wave's stalls short, nothing is issued in a burst: the 8 DMA pieces of a step go out one per MFMA group in
// the second half of the step (each behind 8 MFMAs already queued), the corpus refills in two groups of four.
//
// Same contract as k_scan<256, EPI_EMIT, HAS_MASK, false, false, false, true>: same ScanParams, same tile schedule, same candidate
// segments and counters, emit check of a tile fused into the first k-step of the next one (even number of k-steps: the host checks).
// Vector-memory order of a wave per k-step (the counted waits depend on it): 4 corpus loads (k sub-step 0 registers) | barrier |
// 8 DMA pieces | 4 corpus loads (k sub-step 1 registers) = 16 operations.
//
// Measured (profiles/r04/c4_wave_layout_1x.txt): -7 % at 10 M rows x 1024 queries, -20 % on a 2 M-row shard; LDS waits halved, clock
// 1.60 -> 1.88 GHz, MFMA pipe busy 73 -> 47 % of the cycles. Kept as a measured variant, not as the product path.
#pragma once
#include <utility>

#include "scan_kernel.hpp"

#ifndef W4_EMIT
#define W4_EMIT 1   // 0: timing-only build without the emit path
#endif
#ifndef W4_BURST
#define W4_BURST 0  // 1: the 8 DMA pieces of a step go out together right behind the barrier (as k_scan does) instead of one per MFMA group
#endif
#ifndef W4_PD
#define W4_PD 2     // query-fragment groups read ahead
#endif

namespace rdx {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the MFMA groups of a k-step written out by construction (the unroll
// pragma gives up on a body of this size, and an accumulator array behind a run-time index lives in scratch memory)
template <class F, int... I>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_seq(f, std::make_integer_sequence<int, N>{});
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_keep8(half8 (&a)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 : "n"(N)
                 : "memory");
}

// The MFMAs are inline asm with the accumulator PINNED to the accumulator half of the register file ("a") and tied to itself. With
// 256 accumulators the file is exactly full, and the compiler's own allocation of the builtin answers that by rotating accumulators
// through arch VGPRs (v_accvgpr_read + s_nop 7 behind every MFMA of the hot loop, seen in the listing). Hazards the compiler no longer
// covers, by construction: an accumulator is read by a vector instruction (the emit check) or written again only >= 56 MFMAs after
// the MFMA that wrote it (blocks are visited round-robin, 8 groups of 8 MFMAs per k sub-step); the A/B operands come from memory
// instructions whose completion is waited for (s_waitcnt), never from a vector ALU instruction right in front.
#ifndef W4_BUILTIN
#define W4_BUILTIN 0   // 1 (debug): the compiler's builtin instead — slow (accumulator traffic), but hazard-free by the compiler's own rules: the
                       // reference build that told a scheduling bug of the asm form (vector zeroing of accumulators) from a logic bug
#endif

__device__ __forceinline__ void mfma_acc(f32x4& c, const half8& a, const half8& b) {
#if W4_BUILTIN
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
}
// first MFMA of a tile in this accumulator: C = 0 (the old value is dead: nothing to zero)
__device__ __forceinline__ void mfma_new(f32x4& c, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

template <bool HAS_MASK>
__global__ __launch_bounds__(256) void k_scan_w4(const ScanParams p) {
    constexpr int BN = 256;
    constexpr int B_BYTES = BN * BK * 2;   // one k-step image of the workgroup's 256 queries: 32 KiB
    constexpr int NPB = 8;                 // 1 KiB DMA pieces per wave and image
    constexpr int NA = 8;                  // corpus loads per wave and k-step
    constexpr int V = NA + NPB;            // vector-memory operations per wave and k-step
    constexpr int NB16 = BN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- stream / query tile of this workgroup: as k_scan ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int wpx = gridDim.x >> 3;
    const int G = wpx / p.nqt;
    if (slot >= G * p.nqt) return;
    const int qt = slot % p.nqt;
    const int stream = xcd * G + slot / p.nqt;
    const int n_streams = 8 * G;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..3: rows 64 wave .. 64 wave + 63 of every tile

    const int n_sched = (int)p.n_tiles;                                  // (main pass: tile_stride 1)
    int my_tiles = stream < n_sched ? (n_sched - stream + n_streams - 1) / n_streams : 0;
    int tail_first = 0, bulk_it = my_tiles;
    if (p.use_xlo) {
        const int ls = slot / p.nqt, cnt = p.xlo[xcd + 1] - p.xlo[xcd];
        bulk_it = p.bulk_it;
        tail_first = p.xlo[xcd] + ls;
        my_tiles = bulk_it + (cnt > ls ? (cnt - ls + G - 1) / G : 0);
    }
    auto sched_of = [&](int it_i) __attribute__((always_inline)) { return it_i < bulk_it ? stream + it_i * n_streams : tail_first + (it_i - bulk_it) * G; };
    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x] = wall_clock64();
    const int KS = p.ksteps;
    const int total = my_tiles * KS;

    uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + RING_SLOTS * B_BYTES);   // [256] hit counters of this (stream, query tile)
    float* tau_s = reinterpret_cast<float*>(lcnt + BN);                          // [256] thresholds, stored [l15][block]
    for (int i = threadIdx.x; i < BN; i += 256) {
        lcnt[i] = 0;
        tau_s[(i & 15) * NB16 + (i >> 4)] = p.tau[qt * BN + i];
    }

    const char* qbase = reinterpret_cast<const char*>(p.qshadow) + ((int64_t)qt * KS) * KSTEP_BYTES + wave * (NPB * 1024) + lane * 16;
    auto issue_piece = [&](int ks_i, int slot_i, int i) __attribute__((always_inline)) {
        const char* src = qbase + (int64_t)ks_i * KSTEP_BYTES + i * 1024;
        char* dst = smem + slot_i * B_BYTES + wave * (NPB * 1024) + i * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const int64_t rb_bytes = (int64_t)KS * 4096;   // bytes of one 32-row block in the scan copy
    const uint32_t lane16 = (uint32_t)lane * 16;
    auto a_src = [&](int it_i, int ks_i) __attribute__((always_inline)) -> const char* {   // this wave's FIRST block; the second one follows it
        const int64_t tile = (int64_t)sched_of(it_i);
        return reinterpret_cast<const char*>(p.shadow) + (tile * 8 + wave * 2) * rb_bytes + (int64_t)ks_i * 4096;
    };

    const int l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[4][NB16];   // [16-row block m of the wave's 64 rows][16-query block]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NB16; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;

    const int b_sw = (l15 >> 1) & 7;
    int b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[kk] = l15 * 128 + (((kk * 4 + lq) ^ b_sw) << 4);

    // which of the wave's 64 rows of schedule entry it_done may be used: two 32-bit words (scalar values)
    auto tile_rows = [&](int it_done, int64_t& row_b, uint32_t& ok_lo, uint32_t& ok_hi, bool& filt) __attribute__((always_inline)) {
        const int64_t tile = (int64_t)sched_of(it_done);
        row_b = tile * TILE_ROWS + wave * 64;
        const bool ragged = tile * TILE_ROWS + TILE_ROWS > p.rows;
        ok_lo = ok_hi = 0xffffffffu;
        if (ragged) {
            const int64_t left = p.rows - row_b;
            ok_lo = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
            ok_hi = left >= 64 ? 0xffffffffu : (left <= 32 ? 0u : ((1u << (left - 32)) - 1u));
        }
        if constexpr (HAS_MASK) {
            if (row_b < p.rows) ok_lo &= p.allow[row_b >> 5];
            if (row_b + 32 < p.rows) ok_hi &= p.allow[(row_b >> 5) + 1];
        }
        filt = HAS_MASK || ragged;
    };
    auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    };
    auto block_max = [&](int n) __attribute__((always_inline)) {   // over the 64 rows x 16 queries of query block n: 16 values per lane
        float t = max3(acc[0][n][0], acc[0][n][1], acc[0][n][2]);
        t = max3(t, acc[0][n][3], acc[1][n][0]);
        t = max3(t, acc[1][n][1], acc[1][n][2]);
        t = max3(t, acc[1][n][3], acc[2][n][0]);
        t = max3(t, acc[2][n][1], acc[2][n][2]);
        t = max3(t, acc[2][n][3], acc[3][n][0]);
        t = max3(t, acc[3][n][1], acc[3][n][2]);
        return max3(t, acc[3][n][3], acc[3][n][3]);
    };
    uint2* const cand_s = p.cand;
    const uint32_t capw_s = p.capw;
    // EMIT, rare path (a handful of blocks per tile): the hits of query block n go to the (query, stream) segments. The lane's 16 values
    // are parked in LDS and walked by a ROLLED loop: written out 16 times per call site (32 call sites), the conditional code made
    // the register allocator move accumulators to scratch memory at the top of every tile.
    float* const stage = reinterpret_cast<float*>(tau_s + BN) + (wave * 64 + lane) * 16;   // [wave][lane][16]
    auto emit_block = [&](int n, float tq, int it_done) __attribute__((always_inline)) {
        int64_t row_b;
        uint32_t ok_lo, ok_hi;
        bool filt;
        tile_rows(it_done, row_b, ok_lo, ok_hi, filt);
        int lc = l15, lr = lq * 4;
        asm volatile("" : "+v"(lc), "+v"(lr));   // (not loop invariants worth registers: see k_scan)
        const int ql = n * 16 + lc;
        const uint32_t seg0 = ((uint32_t)(qt * BN + ql) * (uint32_t)n_streams + (uint32_t)stream) * capw_s;
        const uint32_t row0 = (uint32_t)row_b + (uint32_t)lr;
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4*>(stage + 4 * m) = acc[m][n];
        const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&lcnt[ql]);
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const float v = stage[i];
            const int bit = (i >> 2) * 16 + (i & 3) + lr;     // row of the wave's 64: 16 m + 4 lq + r
            const uint32_t okw = bit < 32 ? ok_lo : ok_hi;
            if (v >= tq && (!filt || ((okw >> (bit & 31)) & 1u))) {
                uint32_t pos;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(lds_addr), "v"(1u) : "memory");
                if (pos < capw_s) cand_s[seg0 + pos] = make_uint2(__float_as_uint(v * p.inv_scale2), row0 + (uint32_t)((i >> 2) * 16 + (i & 3)));
            }
        }
    };
    auto tau_one = [&](int tb, int n) __attribute__((always_inline)) { return tau_s[tb + n]; };

    if (total > 0) {
        half8 a0[8], a1[8];   // [block b of the wave's two][1 KiB chunk c = 2 kk + mm]: index 4 b + c
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < NPB; ++i) issue_piece(j % KS, j, i);
        auto first_load = [&](half8 (&af)[8], int j) __attribute__((always_inline)) {
            const bool have = j < total;
            const char* sj = uniform_ptr(a_src(have ? j / KS : 0, have ? j % KS : 0));
            const char* sj2 = uniform_ptr(sj + rb_bytes);
            gload16<false, 0>(af[0], sj, lane16);
            gload16<false, 1024>(af[1], sj, lane16);
            gload16<false, 0>(af[4], sj2, lane16);
            gload16<false, 1024>(af[5], sj2, lane16);
            gload16<false, 2048>(af[2], sj, lane16);
            gload16<false, 3072>(af[3], sj, lane16);
            gload16<false, 2048>(af[6], sj2, lane16);
            gload16<false, 3072>(af[7], sj2, lane16);
        };
        first_load(a0, 0);
        first_load(a1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the only barrier that waits for memory

        int it = 0, ks = 0;
        int it2 = 2 / KS, ks2 = 2 % KS;    // step s + 2 (its corpus fragments are fetched during step s)
        const char* a_next = a_src(it2, ks2);
        const char* const a_first = a_src(0, 0);
        int slot_c = 0;
        int ksb = 3 % KS;

        constexpr int GB = 2;              // query blocks per group: 8 MFMAs
        constexpr int GPK = NB16 / GB;     // 8 groups per k sub-step
        constexpr int NG = 2 * GPK;        // 16 groups per step
        constexpr int QB_BYTES = 2048;
        constexpr int PD = W4_PD;          // groups read ahead (8 VGPRs each): 15 MFMAs between a read and its first use
        constexpr int NBUF = 4;
        constexpr int BAR_G = NG / 2;      // the step's barrier sits in front of this group
        static_assert(BAR_G <= NG - PD, "the read-ahead of the next step's first groups must lie behind the barrier");
        half8 bf[NBUF][GB];
        auto load_group = [&](const char* img, int g, half8 (&dst)[GB]) __attribute__((always_inline)) {
            const int kk = g / GPK, nb0 = (g % GPK) * GB;
#pragma unroll
            for (int j = 0; j < GB; ++j) dst[j] = *reinterpret_cast<const half8*>(img + b_off[kk] + (nb0 + j) * QB_BYTES);
        };
#pragma unroll
        for (int g = 0; g < PD; ++g) load_group(smem, g, bf[g]);

        auto step = [&](auto fuse_tag, half8 (&af)[8], int s, int it_prev) __attribute__((always_inline)) {
            constexpr bool FUSE = decltype(fuse_tag)::value;
            // this step's corpus fragments were issued two steps ago; the 16 operations of the step in between stay in flight
            wait_vmcnt_keep8<V>(af);
            const char* st = smem + slot_c * B_BYTES;
            const char* stn = smem + ((slot_c + 1) & 3) * B_BYTES;
            const bool more = s + 2 < total;
            const char* an = uniform_ptr(more ? a_next : a_first);   // (wave-uniform by construction; the "s" operands below need the compiler to know)
            const char* an2 = uniform_ptr(an + rb_bytes);
            float tq_cur = 0.f;
            int tb = l15 * NB16;
            if constexpr (FUSE) {
                asm volatile("" : "+v"(tb));
                tq_cur = tau_one(tb, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<NG>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                constexpr int kk = g / GPK, nb0 = (g % GPK) * GB;
                if constexpr (g == BAR_G) {
                    // image s+1: my pieces went out during step s-2; since then 4 (its last corpus loads) + 16 (step s-1) + 4 (the
                    // first corpus loads of this step) operations. Behind the barrier every wave has left step s-1: its slot is refilled.
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + V) : "memory");
                    __builtin_amdgcn_s_barrier();
                    if (W4_BURST) {
#pragma unroll
                        for (int i = 0; i < NPB; ++i) issue_piece(ksb, (slot_c + 3) & 3, i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int n = nb0 + j;
                    if constexpr (FUSE) {
                        if (kk == 0) {
                            const float mx = block_max(n);
                            const float tq = tq_cur;
                            if (n + 1 < NB16) tq_cur = tau_one(tb, n + 1);
                            if (W4_EMIT && it_prev >= 0 && __any(mx >= tq)) emit_block(n, tq, it_prev);
                            if (!W4_EMIT) asm volatile("" ::"v"(mx));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const half8& av = af[(m >> 1) * 4 + 2 * kk + (m & 1)];
                        // (first k-step of a tile: C = 0 in the instruction. Zeroing the accumulators with vector writes in front is
                        //  what the compiler schedules freely around the asm MFMAs — measured: stale sums in 12 of 16 query blocks)
                        if (FUSE && kk == 0) mfma_new(acc[m][n], av, bf[g % NBUF][j]);
                        else mfma_acc(acc[m][n], av, bf[g % NBUF][j]);
                        if (m == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (j == 0) {
                                if (g + PD < NG) load_group(st, g + PD, bf[(g + PD) % NBUF]);
                                else load_group(stn, g + PD - NG, bf[(g + PD) % NBUF]);   // first groups of step s+1 (behind the barrier)
                            } else if (!W4_BURST && g >= BAR_G) {
                                issue_piece(ksb, (slot_c + 3) & 3, g - BAR_G);           // image s+3, one piece per group
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if ((g % GPK) == GPK - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    // the matrix pipe has read this sub-step's fragments: refill them with those of step s+2
                    if (kk == 0) {
                        gload16<false, 0>(af[0], an, lane16);
                        gload16<false, 1024>(af[1], an, lane16);
                        gload16<false, 0>(af[4], an2, lane16);
                        gload16<false, 1024>(af[5], an2, lane16);
                    } else {
                        gload16<false, 2048>(af[2], an, lane16);
                        gload16<false, 3072>(af[3], an, lane16);
                        gload16<false, 2048>(af[6], an2, lane16);
                        gload16<false, 3072>(af[7], an2, lane16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };

        auto epilogue = [&](int it_done) __attribute__((always_inline)) {   // the stream's last tile
            int tb = l15 * NB16;
            asm volatile("" : "+v"(tb));
#pragma unroll
            for (int n = 0; n < NB16; ++n) {
                const float mx = block_max(n);
                const float tq = tau_one(tb, n);
                if (W4_EMIT && __any(mx >= tq)) emit_block(n, tq, it_done);
                if (!W4_EMIT) asm volatile("" ::"v"(mx));
            }
        };

        auto advance = [&]() __attribute__((always_inline)) {
            if (++ks == KS) { ks = 0; ++it; }
            if (++ks2 == KS) {
                ks2 = 0;
                ++it2;
                a_next = a_src(it2, 0);   // (beyond the stream's last tile: an address nobody loads from)
            } else {
                a_next += 4096;
            }
            slot_c = (slot_c + 1) & 3;
            if (++ksb == KS) ksb = 0;
        };

        int s = 0;
        auto pair = [&](auto first_tag, int it_prev) __attribute__((always_inline)) {
            step(first_tag, a0, s, it_prev);
            advance();
            step(std::false_type{}, a1, s + 1, 0);
            advance();
            s += 2;
        };
        // (the first tile goes through the checking step too, with nothing to check: it_prev = -1. A peeled first tile is two more
        //  copies of the step, and the register assignment of 256 + 256 registers has no room for the moves between copies)
        for (int t = 0; t < my_tiles; ++t) {
            pair(std::true_type{}, t - 1);
            for (int j = 2; j < KS; j += 2) pair(std::false_type{}, 0);
        }
        epilogue(my_tiles - 1);
        // drain the never-consumed tail prefetches; naming all fragments keeps their registers reserved until here
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a0[4]), "v"(a0[5]), "v"(a0[6]), "v"(a0[7]),
                     "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a1[4]), "v"(a1[5]), "v"(a1[6]), "v"(a1[7])
                     : "memory");
    }

    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x + 1] = wall_clock64();
    __syncthreads();
    for (int i = threadIdx.x; i < BN; i += 256) p.cntw[(int64_t)(qt * BN + i) * n_streams + stream] = lcnt[i];
}

}  // namespace rdx
