// rdx_common.hpp — shared device helpers and the HBM layouts of librdx (gfx950 only).
//
// HBM layout of one corpus shard (all owned by rdx_index, see rdx_api.hip):
//   master  fp32 [cap_rows][dim]   row-major L2-normalised rows; the exact re-score and the exact scan read it
//   shadow  fp16 "scan copy", value = master * 2^scale_log2, stored in MFMA FRAGMENT ORDER:
//           [row block rb = row/32][k chunk c = k/32][row half m][lane 0..63][8 halfs]
//           lane l of chunk (rb, c, m) holds row rb*32 + m*16 + (l & 15), k = c*32 + 8*(l >> 4) + 0..7 — exactly the A
//           operand of v_mfma_f32_16x16x32_f16 (same cycles per flop as the 32x32x16 shape, but the chip holds a higher clock
//           under it, MI355X_MICROARCH.md "DVFS give-back" item 7: +4.5 % on the B = 1024 scan, measured in round 1).
//           One chunk = 1 KiB = one fully coalesced global_load_dwordx4 of a wavefront, and all chunks of a 32-row block
//           are contiguous (dim_pad/16 KiB): a wave streams its rows straight from HBM into VGPRs, no LDS, no address
//           arithmetic beyond "+1 KiB".
//   query scan copy (per search): [query block of 256][k-step ks = k/64][256 rows][64 k] fp16, i.e. one 32 KiB LDS image
//           per (block, k-step). Inside an image row r (128 B = eight 16-B chunks) chunk c sits in slot c ^ ((r >> 1) & 7):
//           the XOR makes the ds_read_b128 fragment reads of 16 different rows hit 16 different 16-B bank slots
//           (conflict-free; cdna_hip_programming.md T2 / rule 21: swizzle where the image is written and on the read).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 256;                    // corpus rows per scan tile / shadow block
constexpr int BK = 64;                            // k elements per k-step image
constexpr int KSTEP_BYTES = TILE_ROWS * BK * 2;   // 32 KiB
constexpr int MAX_DIM = 4096;

// offset (in halfs) of element (row r, column k) inside the fragment-ordered corpus scan copy
__host__ __device__ inline int64_t corpus_off(int64_t r, int k, int ksteps) {
    // 16x16x32 A operand: chunk (rb, c = k/32, m = row half): lane l = row m*16 + (l & 15), k = c*32 + 8*(l >> 4) + 0..7
    const int64_t rb16 = r >> 5;
    const int m = (int)((r >> 4) & 1), c = k >> 5;
    const int lane16 = (int)(r & 15) + (((k & 31) >> 3) << 4);
    return (((rb16 * (ksteps * 2) + c) * 2 + m) * 64 + lane16) * 8 + (k & 7);
}

// offset (in halfs) of element (query r, column k) inside the tiled + swizzled query scan copy
__host__ __device__ inline int64_t query_off(int64_t r, int k, int ksteps) {
    const int64_t blk = r >> 8;
    const int rr = (int)(r & 255);
    const int ks = k >> 6;
    const int chunk = (k & 63) >> 3;
    const int slot = chunk ^ ((rr >> 1) & 7);
    return ((blk * ksteps + ks) * 256 + rr) * 64 + slot * 8 + (k & 7);
}

// "lane order" reduction shared with oracle/rdx_oracle.c: butterfly p[l] += p[l ^ m], m = 32..1
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// monotone map float -> uint32 (ascending), used by every radix select
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th largest key among the keys a block holds (1 <= k <= number of keys) by MSB-first 8-bit radix passes; all threads
// of the block call it. `scan(f)` calls f(key) for every key THIS thread owns (from global memory, LDS or registers).
// hist: HIST_WORDS words of LDS, bc: 4 words of LDS. Returns the key; *n_gt = #keys strictly greater.
// The histogram is kept in HIST_COPIES interleaved copies, lane l adds to copy l % HIST_COPIES: the digits of a score row are
// heavily skewed (the top byte of most keys is one of 3-5 values), and 64 lanes adding to one LDS address are served one
// after the other; with 8 copies at most 8 queue on an address.
constexpr int HIST_COPIES = 8;
constexpr int HIST_WORDS = 256 * HIST_COPIES;
template <class Scan>
__device__ uint32_t block_kth_largest_scan(Scan scan, int64_t k, uint32_t* hist, uint32_t* bc, int64_t* n_gt) {
    uint32_t prefix = 0, pmask = 0;
    int64_t remaining = k, gt = 0;
    for (int i = threadIdx.x; i < HIST_WORDS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t my_copy = threadIdx.x & (HIST_COPIES - 1);
    // two workgroup barriers per pass: wave 0 clears each bin right after reading it (the histogram is ready for the next
    // pass), and nobody overwrites bc[] before the next pass's first barrier
    for (int shift = 24; shift >= 0; shift -= 8) {
        // (measured and dropped: counting the skewed first digit with wave ballots instead of per-key atomics doubles the
        //  time of the four passes — the ballot / shuffle loop costs more than the queued LDS adds)
        scan([&](uint32_t key) {
            if ((key & pmask) == prefix) atomicAdd(&hist[(((key >> shift) & 255u) << 3) | my_copy], 1u);
        });
        __syncthreads();
        // bin where the count taken from the top reaches `remaining`: wave 0, lane l owns bins 4l..4l+3, suffix sums by
        // shuffles, crossing lane by ballot (the counts are < 2^32 here; remaining <= n)
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            uint32_t hb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {   // bins 4l..4l+3: sum their copies (two 16-byte reads each) and clear them for the next pass
                uint4* cp = reinterpret_cast<uint4*>(hist + (4 * l + b) * HIST_COPIES);
                const uint4 a = cp[0], c = cp[1];
                hb[b] = a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
                cp[0] = make_uint4(0, 0, 0, 0);
                cp[1] = make_uint4(0, 0, 0, 0);
            }
            const uint32_t h0 = hb[0], h1 = hb[1], h2 = hb[2], h3 = hb[3];
            uint64_t suf = (uint64_t)h0 + h1 + h2 + h3;   // becomes sum over bins >= 4l
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint64_t v = __shfl_down(suf, off, 64);
                if (l + off < 64) suf += v;
            }
            const unsigned long long m = __ballot(suf >= (uint64_t)remaining);   // never empty: suf(lane 0) = n' >= remaining
            const int L = 63 - __clzll(m);
            if (l == L) {
                uint32_t acc = (uint32_t)(suf - ((uint64_t)h0 + h1 + h2 + h3));   // count in bins above this lane's
                int b = 4 * l + 3;
                const uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int j = 3; j > 0; --j) {
                    if ((uint64_t)acc + hh[j] >= (uint64_t)remaining) break;
                    acc += hh[j];
                    --b;
                }
                bc[0] = (uint32_t)b;
                bc[1] = acc;
            }
        }
        __syncthreads();
        prefix |= bc[0] << shift;
        pmask |= 255u << shift;
        remaining -= bc[1];
        gt += bc[1];
    }
    __syncthreads();
    *n_gt = gt;
    return prefix;
}

// the same over n keys addressed by index (key_at(i), i < n): thread t takes i = t, t + blockDim, ...
template <class KeyAt>
__device__ uint32_t block_kth_largest(KeyAt key_at, int64_t n, int64_t k, uint32_t* hist, uint32_t* bc, int64_t* n_gt) {
    return block_kth_largest_scan(
        [&](auto f) {
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) f(key_at(i));
        },
        k, hist, bc, n_gt);
}

}  // namespace rdx
