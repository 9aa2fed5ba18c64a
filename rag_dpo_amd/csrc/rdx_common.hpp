// rdx_common.hpp — shared device helpers and the HBM layouts of librdx (gfx950 only).
//
// HBM layout of one corpus shard (all owned by rdx_index, see rdx_api.hip):
//   master  fp32 [cap_rows][dim]            row-major L2-normalised rows; the exact re-score reads it
//   shadow  fp16 [cap_rows/256][dim_pad/64][256 rows][64 k]   "scan copy": value = master * 2^scale_log2,
//           one 32 KiB image per (256-row tile, 64-wide k-step), stored in the exact byte order the MFMA
//           scan kernel wants in LDS, so a k-step is ONE contiguous 32 KiB stream from HBM and the
//           global->LDS DMA (global_load_lds_dwordx4) needs no per-lane address arithmetic.
//           Inside an image row r (128 B = eight 16-B chunks) chunk c sits in slot c ^ ((r >> 1) & 7):
//           the XOR makes the ds_read_b128 fragment reads of 16 different rows hit 16 different 16-B
//           bank slots (conflict-free), cf. cdna_hip_programming.md T2 / rule 21 (swizzle is applied
//           where the image is WRITTEN, i.e. once at ingest, and again on the LDS read address).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE_ROWS = 256;                    // corpus rows per scan tile / shadow block
constexpr int BK = 64;                            // k elements per k-step image
constexpr int KSTEP_BYTES = TILE_ROWS * BK * 2;   // 32 KiB
constexpr int MAX_DIM = 4096;
constexpr int SETS_PER_WAVE = 32;                 // threshold-bootstrap sets per (stream, wave row)

// offset (in halfs) of element (row r, column k) inside the tiled fp16 copy
__host__ __device__ inline int64_t shadow_off(int64_t r, int k, int ksteps) {
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

// k-th largest key among n keys (1 <= k <= n) by MSB-first 8-bit radix passes; all threads of the block
// call it. hist: 256 words of LDS, bc: 4 words of LDS. Returns the key; *n_gt = #keys strictly greater.
template <class KeyAt>
__device__ uint32_t block_kth_largest(KeyAt key_at, int64_t n, int64_t k, uint32_t* hist, uint32_t* bc,
                                      int64_t* n_gt) {
    uint32_t prefix = 0, pmask = 0;
    int64_t remaining = k, gt = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t key = key_at(i);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            int b = 255;
            for (; b > 0; --b) {
                if ((int64_t)acc + hist[b] >= remaining) break;
                acc += hist[b];
            }
            bc[0] = (uint32_t)b;
            bc[1] = acc;
        }
        __syncthreads();
        prefix |= bc[0] << shift;
        pmask |= 255u << shift;
        remaining -= bc[1];
        gt += bc[1];
        __syncthreads();
    }
    *n_gt = gt;
    return prefix;
}

}  // namespace rdx
