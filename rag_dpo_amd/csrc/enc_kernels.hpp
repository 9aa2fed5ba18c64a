// enc_kernels.hpp — E1/E2: the two memory-bound pieces of the query encoder's forward that PyTorch runs as several kernels each.
// (`SentenceTransformer.encode`, reference src/utils/embedding_provider.py:139-145; SURVEY.md §8 row f2. The GEMMs stay hipBLASLt
// through torch: rag_dpo_amd/embedding_provider.py `_PackedEncoder`.)
//
// E1 k_enc_attention: self-attention of SHORT texts straight on the packed QKV projection. A batch of questions is ~20 tokens
//    per text: the score matrix of a (text, head) is 20 x 20, the arithmetic is nothing, and what the padded-batch form costs is
//    memory passes — scatter the packed projection into [batch][longest] slots, attention over the padding, a transposing copy,
//    gather back (~0.7 GB per layer for BASELINE config 5's 1024 questions). Here: reads qkv [T][3H] once (keys and values of a
//    text again from L1/L2 for each of its tokens), writes ctx [T][H] once. Roofline: HBM, T*4H*2 B per launch.
//    Layout: 4 adjacent lanes own one (token, head) row, 16 of its 64 dimensions each: the row's q slice lives in registers, a key's
//    score is 8 v_dot2_f32_f16 + a sum over the quad (DPP), soft-max is the running-max form in fp32, the lane accumulates its 16
//    output dimensions (v_fma_mix_f32 straight from the fp16 values). The keys and values of a workgroup's texts are staged in LDS.
//    A quad's lanes share the text, so they run the same number of keys; quads of one wave may not.
// E2 k_enc_add_ln: y = LayerNorm(a + b) * gamma + beta over rows of `hidden` halves, one wave per row, values kept in registers
//    between the statistics and the output (torch: an add kernel, then LayerNorm: five passes over the row instead of three).
//    The sum is rounded to fp16 before the statistics, as the two-kernel form does.
#pragma once
#include <hip/hip_fp16.h>

#include "rdx_common.hpp"

namespace rdx {

constexpr int ENC_HEAD_DIM = 64;

template <int CTRL>
__device__ __forceinline__ float quad_swap_add(float v) {   // v + (the value of the lane CTRL's quad permutation names)
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(o);
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// o + p * half(v), the half taken from the low / high 16 bits of a packed pair: ONE instruction (the compiler's choice for
// `o += p * (float)v` is a v_cvt_f32_f16 per element plus packed fp32 FMAs: 24 instead of 16 instructions per key)
__device__ __forceinline__ float fma_mix_lo(float p, uint32_t v, float o) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(p), "v"(v), "v"(o));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(float p, uint32_t v, float o) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(p), "v"(v), "v"(o));
    return r;
}

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dot16(const h8& q0, const h8& q1, const h8& k0, const h8& k1) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s = __builtin_amdgcn_fdot2(h2{q0[2 * e], q0[2 * e + 1]}, h2{k0[2 * e], k0[2 * e + 1]}, s, false);
        s = __builtin_amdgcn_fdot2(h2{q1[2 * e], q1[2 * e + 1]}, h2{k1[2 * e], k1[2 * e + 1]}, s, false);
    }
    s = quad_swap_add<0xB1>(s);   // quad_perm [1,0,3,2]
    return quad_swap_add<0x4E>(s);   // quad_perm [2,3,0,1]
}

constexpr int ENC_KEY_WINDOW = 192;   // most keys (and values) of one head a workgroup stages in LDS: 64 rows of texts up to 64 tokens need <= 190;
                                      // texts up to L tokens need 64 + 2 (L - 1): the host passes that (less LDS per workgroup = more of them per CU)

// One row's attention over the S keys at `kv` (key j: 16 of its 64 dims at kv + j * STRIDE halves, its value VOFF halves further).
// Two keys per iteration: one running-max update and one rescale of the 16 accumulators for both (an odd text's last iteration
// reads its last key twice and gives the copy the score -inf, i.e. weight 0).
template <class P>
__device__ __forceinline__ void attend_row(P kv, int64_t stride, int64_t voff, int S, const h8& q0, const h8& q1, float sc, float (&o)[16], float& l) {
    float m = -INFINITY;
    for (int j = 0; j < S; j += 2, kv += 2 * stride) {
        const bool two = j + 1 < S;
        P kv2 = two ? kv + stride : kv;
        const h8 ka0 = reinterpret_cast<const h8*>(kv)[0], ka1 = reinterpret_cast<const h8*>(kv)[1];
        const h8 kb0 = reinterpret_cast<const h8*>(kv2)[0], kb1 = reinterpret_cast<const h8*>(kv2)[1];
        const u4 va0 = reinterpret_cast<const u4*>(kv + voff)[0], va1 = reinterpret_cast<const u4*>(kv + voff)[1];
        const u4 vb0 = reinterpret_cast<const u4*>(kv2 + voff)[0], vb1 = reinterpret_cast<const u4*>(kv2 + voff)[1];
        const float sa = dot16(q0, q1, ka0, ka1) * sc;
        const float sb = two ? dot16(q0, q1, kb0, kb1) * sc : -INFINITY;
        const float mn = fmaxf(m, fmaxf(sa, sb));
        const float corr = __builtin_amdgcn_exp2f(m - mn);   // first iteration: exp2(-inf) = 0
        const float pa = __builtin_amdgcn_exp2f(sa - mn), pb = __builtin_amdgcn_exp2f(sb - mn);
        l = l * corr + pa + pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = fma_mix_lo(pb, vb0[e], fma_mix_lo(pa, va0[e], o[2 * e] * corr));
            o[2 * e + 1] = fma_mix_hi(pb, vb0[e], fma_mix_hi(pa, va0[e], o[2 * e + 1] * corr));
            o[8 + 2 * e] = fma_mix_lo(pb, vb1[e], fma_mix_lo(pa, va1[e], o[8 + 2 * e] * corr));
            o[8 + 2 * e + 1] = fma_mix_hi(pb, vb1[e], fma_mix_hi(pa, va1[e], o[8 + 2 * e + 1] * corr));
        }
        m = mn;
    }
}

// grid (ceil(T / 64), heads), 256 threads = 64 consecutive tokens of one head. The texts those tokens belong to are a contiguous
// range of tokens: their keys and values of this head (256 B per token) are staged in LDS once per workgroup — every key is
// wanted by all tokens of its text, and 16-byte loads of the same few lines by every lane are what the CU's vector L1 is slowest
// at (64 B/clk against the LDS's 256 B/clk with broadcast). Rows whose text does not lie inside the window (texts longer than
// the kernel is meant for) read global memory instead: same arithmetic.
__global__ __launch_bounds__(256) void k_enc_attention(const _Float16* __restrict__ qkv, const int32_t* __restrict__ tok_first,
                                                       const int32_t* __restrict__ tok_len, int64_t T, int heads, float scale,
                                                       int window, _Float16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) char enc_smem[];
    _Float16* kv_s = reinterpret_cast<_Float16*>(enc_smem);   // [window][K 64 | V 64]; the host sizes the window from the longest text
    const int g = threadIdx.x & 3;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int64_t t = t0 + (threadIdx.x >> 2);
    const int h = blockIdx.y;
    const int64_t H = (int64_t)heads * ENC_HEAD_DIM, row = 3 * H;
    const bool live = t < T;
    const int64_t tt = live ? t : T - 1;
    const int S = live ? tok_len[tt] : 0;
    const int64_t first = tok_first[tt];
    const int64_t col = (int64_t)h * ENC_HEAD_DIM + g * 16;
    // the window: from the first token of the text of this workgroup's first row
    const int64_t tl = t0 + 63 < T ? t0 + 63 : T - 1;
    const int64_t kfirst = tok_first[t0];
    const int64_t kend = (int64_t)tok_first[tl] + tok_len[tl];
    const int nk = (int)(kend - kfirst < window ? kend - kfirst : window);
    for (int c = threadIdx.x; c < nk * 16; c += 256) {
        const int tok = c >> 4, part = c & 15;   // 16-byte pieces 0..7: the key's 64 dims, 8..15: the value's
        const _Float16* src = qkv + (kfirst + tok) * row + H * (1 + (part >> 3)) + h * ENC_HEAD_DIM + (part & 7) * 8;
        *reinterpret_cast<h8*>(kv_s + tok * 128 + part * 8) = *reinterpret_cast<const h8*>(src);
    }
    const h8* qp = reinterpret_cast<const h8*>(qkv + tt * row + col);
    const h8 q0 = qp[0], q1 = qp[1];
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc = scale * LOG2E;                      // scores in units of log2: exp2 in attend_row
    float l = 0.f;
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    __syncthreads();
    const int64_t rel = first - kfirst;                  // >= 0: texts are stored in order
    if (rel + S <= nk) attend_row(kv_s + rel * 128 + g * 16, (int64_t)128, (int64_t)64, S, q0, q1, sc, o, l);
    else attend_row(qkv + first * row + H + col, row, H, S, q0, q1, sc, o, l);
    if (live) {
        const float inv = 1.f / l;   // S >= 1: l >= 1
        h8 r0, r1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r0[e] = (_Float16)(o[e] * inv);
            r1[e] = (_Float16)(o[8 + e] * inv);
        }
        h8* op = reinterpret_cast<h8*>(ctx + t * H + col);
        op[0] = r0;
        op[1] = r1;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// NCH = hidden / 512 (each lane holds NCH chunks of 8 halves); grid ceil(rows / 4), 256 threads = 4 rows
template <int NCH>
__global__ __launch_bounds__(256) void k_enc_add_ln(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                    const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta, float eps,
                                                    int64_t rows, _Float16* __restrict__ out) {
    constexpr int HID = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float x[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int64_t off = r * HID + c * 512 + lane * 8;
        const h8 va = *reinterpret_cast<const h8*>(a + off), vb = *reinterpret_cast<const h8*>(b + off);
        const h8 vs = va + vb;   // fp16 sum, rounded like the stand-alone add kernel's
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[c][e] = (float)vs[e];
            sum += x[c][e];
        }
    }
    const float mean = wave_sum(sum) * (1.f / HID);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = x[c][e] - mean;
            sq += d * d;
        }
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / HID) + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cc = c * 512 + lane * 8;
        const h8 gw = *reinterpret_cast<const h8*>(gamma + cc), bw = *reinterpret_cast<const h8*>(beta + cc);
        h8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (_Float16)((x[c][e] - mean) * rstd * (float)gw[e] + (float)bw[e]);
        *reinterpret_cast<h8*>(out + r * HID + cc) = y;
    }
}

// E3 k_enc_linear_small: out[T][N] = act(x[T][K] W[N][K]^T + bias) for SMALL token counts (T <= 256: one question, a question's
//    sub-queries) — the shape where the BLAS library's GEMM takes 11 us whatever the size and the work is reading the weight matrix
//    once (2 - 8 MB). One workgroup per 16 output features: its 4 waves split K, every wave streams its quarter of the 16 weight rows
//    straight into MFMA A fragments (128 contiguous bytes per row and k-step of 64), the activations are the B operand (re-read from
//    L2 by every workgroup: T*K*2 B, small), v_mfma_f32_16x16x32_f16 with the token on the lane; the four partial tiles meet in LDS,
//    bias and the optional erf-GELU are applied in fp32, fp16 out. Roofline: HBM, N*K*2 B per launch. Pays up to ~32 tokens (one
//    question): every workgroup re-reads all activations, so from 64 tokens on the BLAS library is as fast or faster and the host
//    (rag_dpo_amd/embedding_provider.py SMALL_TOKENS) calls that instead; correct for any T <= 256.
//    NTB = 16-token blocks (tokens beyond T repeat the last row and are not stored).
//    NW = waves per workgroup that split K (4, or 16 for one or two token blocks: the whole slice of a wave — at most four k-steps of
//    64 — is then requested before the first MFMA, so the kernel pays the memory latency once: 9.3 -> x us per projection of a question).
template <int NTB, bool GELU, int NW>
__global__ __launch_bounds__(NW * 64) void k_enc_linear_small(const _Float16* __restrict__ x, const _Float16* __restrict__ w,
                                                              const _Float16* __restrict__ bias, int T, int N, int K,
                                                              _Float16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    float* slab = reinterpret_cast<float*>(lin_smem);   // [NW waves][NTB * 4][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int l15 = lane & 15, lq = lane >> 4;
    const int kq = K / NW;                               // this wave's slice of K (a multiple of 64)
    const _Float16* wp = w + (int64_t)(n0 + l15) * K + wave * kq + lq * 8;
    const _Float16* xp[NTB];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const int t = tb * 16 + l15;
        xp[tb] = x + (int64_t)(t < T ? t : T - 1) * K + wave * kq + lq * 8;
    }
    f32x4 acc[NTB];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NW == 16) {
        // kq = 64 .. 256: everything in flight at once
        const int steps = kq >> 6;
        half8 a[4][2] = {}, b[NTB][4][2] = {};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < steps) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    a[s][j] = *reinterpret_cast<const half8*>(wp + s * 64 + j * 32);
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb) b[tb][s][j] = *reinterpret_cast<const half8*>(xp[tb] + s * 64 + j * 32);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < steps) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][j], b[tb][s][j], acc[tb], 0, 0, 0);
            }
    } else {
        for (int k0 = 0; k0 < kq; k0 += 128) {           // two k-steps of 64 per iteration: all their loads first
            half8 a[4], b[NTB][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const half8*>(wp + k0 + j * 32);
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[tb][j] = *reinterpret_cast<const half8*>(xp[tb] + k0 + j * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[tb][j], acc[tb], 0, 0, 0);
        }
    }
    // D[feature = lq * 4 + r][token = l15]: park the partial tile, sum the waves', finish
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(wave * NTB * 4 + tb * 4 + r) * 64 + lane] = acc[tb][r];
    __syncthreads();
    if (wave >= 4) return;
    const int r = wave;                                   // thread -> (register r, lane): feature = (lane >> 4) * 4 + r, token = tb * 16 + (lane & 15)
    const int f = n0 + lq * 4 + r;
    const float bv = (float)bias[f];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const int t = tb * 16 + l15;
        float v = bv;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += slab[(wv * NTB * 4 + tb * 4 + r) * 64 + lane];
        if constexpr (GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        if (t < T) out[(int64_t)t * N + f] = (_Float16)v;
    }
}

}  // namespace rdx

namespace rdx {

// E12 k_enc_attention_mfma: self-attention of packed texts of ANY length (the corpus side: the indexer embeds `heading\n\ntext` chunks of
//    up to ~1 K tokens, reference src/processing/create_chromadb_index.py:300-387, src/utils/embedding_provider.py:30-31,136-145) on the
//    matrix cores, flash-style: no padding, no scatter / gather / transpose passes around it, scores never leave the registers.
//    Work unit = (64 consecutive queries of one text, head): 4 waves x 16 queries; the text's keys and values pass through LDS in tiles of
//    32 (two buffers, the next tile's global loads in flight under the current tile's arithmetic, one barrier per tile).
//      S^T = K Q^T   A = a K tile's rows (16 keys x 32 dims per MFMA, ds_read_b128 from a swizzled image), B = the wave's Q fragments
//                    (registers for the whole kernel): the lane holds, for ITS query, the scores of keys 4g + r and 16 + 4g + r.
//      online soft-max per query = per lane: 8 local values + two cross-lane maxima per tile; the running sum stays lane-partial
//                    (the rescale factor is the query's, i.e. equal in the four lanes that share it) and is reduced once at the end.
//      O^T += V^T P^T  P^T (fp16) IS the B operand as it stands (k position 8g + j <-> key 4g + j | 16 + 4g + j - 4); V^T comes out of the
//                    row-major V tile by ds_read_b64_tr_b16 (the hardware's transposed read: lane i of a 16-lane group receives column i
//                    of 4 rows) in exactly that key order; 16-byte chunks of a V row are stored XOR-ed with (key >> 1) & 3 on their
//                    16-dim block index, which makes the 32 eight-byte reads of a half-wave hit 32 different bank pairs.
//    qb[i] = {first token of the text, its length, first query (within the text) of unit i, 0}. Roofline: MFMA for long texts
//    (4 * len^2 * 64 flop per text and head); measured in profiles/r04/ingest*.json.
constexpr int ATT_KT = 32;
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// (five waves per SIMD: the kernel is bound by the latency of a tile's dependent chain, and left alone the compiler takes 98 + 16
//  registers = four waves; it fits 90 without a spill. Six waves spill 37.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_enc_attention_mfma(const _Float16* __restrict__ qkv, const int32_t* __restrict__ qb, int heads, float scale_log2,
                                                            _Float16* __restrict__ ctx) {
    __shared__ __attribute__((aligned(16))) _Float16 ks[2][ATT_KT * 64];
    __shared__ __attribute__((aligned(16))) _Float16 vs[2][ATT_KT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int first = qb[4 * blockIdx.x], len = qb[4 * blockIdx.x + 1], q0 = qb[4 * blockIdx.x + 2];
    const int h = blockIdx.y;
    const int64_t H = (int64_t)heads * 64, row = 3 * H;
    const int qi = q0 + wave * 16 + l15;
    half8 bq[2];
    {
        const _Float16* qp = qkv + (int64_t)(first + (qi < len ? qi : len - 1)) * row + h * 64 + g * 8;
        bq[0] = *reinterpret_cast<const half8*>(qp);
        bq[1] = *reinterpret_cast<const half8*>(qp + 32);
    }
    // staging: thread -> (key tid >> 3 of the tile, 16-byte chunk tid & 7 of its K row and of its V row)
    const int sk = threadIdx.x >> 3, sc = threadIdx.x & 7;
    const int k_slot = (sk * 8 + (sc ^ ((sk >> 1) & 7))) * 8;                                   // halves
    const int v_slot = sk * 64 + ((((sc >> 1) ^ ((sk >> 1) & 3)) << 1) | (sc & 1)) * 8;
    const _Float16* kv0 = qkv + (int64_t)first * row + H + h * 64 + sc * 8;
    // two tiles travel in registers (sets a and b): the tile after next is requested while this one is multiplied — one tile's arithmetic
    // (~300 cycles) does not cover a global load's ~2 000
    half8 kra, vra, krb, vrb;
    auto load_tile = [&](int t, half8& kr, half8& vr) __attribute__((always_inline)) {
        const int key = t * ATT_KT + sk;
        if (key < len) {
            const _Float16* p = kv0 + (int64_t)key * row;
            kr = *reinterpret_cast<const half8*>(p);
            vr = *reinterpret_cast<const half8*>(p + H);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) kr[e] = vr[e] = (_Float16)0.f;   // (a masked key has weight 0: its value must be finite)
        }
    };
    const int ntiles = (len + ATT_KT - 1) / ATT_KT;
    load_tile(0, kra, vra);
    *reinterpret_cast<half8*>(&ks[0][k_slot]) = kra;
    *reinterpret_cast<half8*>(&vs[0][v_slot]) = vra;
    if (ntiles > 1) load_tile(1, kra, vra);
    if (ntiles > 2) load_tile(2, krb, vrb);
    __syncthreads();
    float m = -INFINITY, lpart = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // addresses of this lane's reads inside a tile (halves)
    int ka[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = kt * 16 + l15;
            ka[kt][s] = (r * 8 + ((s * 4 + g) ^ ((r >> 1) & 7))) * 8;
        }
    const int tq = l15 >> 2, tp = l15 & 3;   // transposed read: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3 of the block
    int va[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 16 + 4 * g + tq;
        va[j] = r * 64 + tp * 4;              // + the swizzled 16-dim block, per dt, below
    }
    const int vsw0 = ((4 * g + tq) >> 1) & 3, vsw1 = ((16 + 4 * g + tq) >> 1) & 3;
    auto tile_step = [&](int t, int buf, half8& kr, half8& vr) __attribute__((always_inline)) {   // kr / vr: tile t + 1 (this step's register set)
        f32x4 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&ks[buf][ka[kt][0]]), bq[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&ks[buf][ka[kt][1]]), bq[1], st[kt], 0, 0, 0);
        }
        float sc8[2][4];
        float mt = -INFINITY;
        const int kbase = t * ATT_KT + 4 * g;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc8[kt][r] = (kbase + kt * 16 + r < len) ? st[kt][r] * scale_log2 : -INFINITY;
                mt = fmaxf(mt, sc8[kt][r]);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);               // finite: every tile holds at least one real key
        const float corr = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float psum = 0.f;
        half8 pb;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc8[kt][r] - mn);
                psum += p;
                pb[kt * 4 + r] = (_Float16)p;
            }
        lpart = lpart * corr + psum;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] *= corr;
            const fp16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(&vs[buf][va[0] + ((dt ^ vsw0) << 4)]));
            const fp16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(&vs[buf][va[1] + ((dt ^ vsw1) << 4)]));
            const half8 av = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3], (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, pb, o[dt], 0, 0, 0);
        }
        if (t + 1 < ntiles) {
            *reinterpret_cast<half8*>(&ks[buf ^ 1][k_slot]) = kr;
            *reinterpret_cast<half8*>(&vs[buf ^ 1][v_slot]) = vr;
            if (t + 3 < ntiles) load_tile(t + 3, kr, vr);     // the set is free again: the tile three ahead
        }
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile_step(t, 0, kra, vra);
        if (t + 1 < ntiles) tile_step(t + 1, 1, krb, vrb);
    }
    float l = lpart;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qi < len) {
        const float inv = 1.f / l;
        _Float16* op = ctx + (int64_t)(first + qi) * H + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            half4 r4;
#pragma unroll
            for (int r = 0; r < 4; ++r) r4[r] = (_Float16)(o[dt][r] * inv);
            *reinterpret_cast<half4*>(op + dt * 16) = r4;
        }
    }
}

// E13 k_enc_gelu: the erf GELU of an fp16 activation IN PLACE. The FFN's first projection is a library GEMM (bias in its epilogue); its
// [T][4096] output (168 MB at 20 K tokens) then took the framework's GELU operation: 101 us of a 248 us FFN1
// (profiles/r04/blas_gemm_layouts.txt), and not for the bytes (3.3 TB/s): libm's erff is ~40 vector instructions per element. Here
// erf comes from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: one reciprocal, one exp2, a degree-5 polynomial):
//   z = x / sqrt 2, t = 1 / (1 + p |z|), q = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) e^(-z^2) = erfc(|z|)
//   gelu(x) = x/2 (1 + erf z) = x/2 (2 - q)  for z >= 0,  x/2 q  for z < 0
// fp32 arithmetic, one rounding to fp16. Against torch.nn.functional.gelu on fp16: the same fp16 value or its neighbour wherever
// |gelu| >= 6e-5, within 1e-6 absolutely in the underflowing negative tail (tests/test_embedding_provider.py), NaN -> NaN.
// HBM-bound target: 4 B per element. grid = min(n / 2048, 2048) workgroups of 256 lanes, 16 B per lane per step.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_enc_gelu(_Float16* __restrict__ x, int64_t n8) {
    // two elements per vector instruction where the ISA has one (v_pk_fma_f32 / v_pk_mul_f32): the kernel is bound by the vector ALU
    const f32x2 c5 = {1.061405429f, 1.061405429f}, c4 = {-1.453152027f, -1.453152027f}, c3 = {1.421413741f, 1.421413741f},
                c2 = {-0.284496736f, -0.284496736f}, c1 = {0.254829592f, 0.254829592f}, one = {1.f, 1.f}, pp = {0.3275911f, 0.3275911f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        half8 v = *reinterpret_cast<const half8*>(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 f = {(float)v[j], (float)v[j + 1]};
            const f32x2 h = 0.5f * f, z = 0.70710678118654752f * f;
            const f32x2 az = {__builtin_fabsf(z[0]), __builtin_fabsf(z[1])};
            const f32x2 d = __builtin_elementwise_fma(pp, az, one);
            const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            f32x2 pl = __builtin_elementwise_fma(c5, t, c4);
            pl = __builtin_elementwise_fma(pl, t, c3);
            pl = __builtin_elementwise_fma(pl, t, c2);
            pl = __builtin_elementwise_fma(pl, t, c1);
            const f32x2 e2 = -1.4426950408889634f * z * z;
            const f32x2 ex = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};   // exp2 of -inf = 0
            const f32x2 q = pl * t * ex;                                                        // erfc(|z|)
            const f32x2 w = {z[0] >= 0.f ? 2.0f - q[0] : q[0], z[1] >= 0.f ? 2.0f - q[1] : q[1]};   // (NaN: the comparison is false -> NaN * NaN)
            const f32x2 r = h * w;
            v[j] = (_Float16)r[0];
            v[j + 1] = (_Float16)r[1];
        }
        *reinterpret_cast<half8*>(x + i * 8) = v;
    }
}

}  // namespace rdx
