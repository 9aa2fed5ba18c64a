// enc_small.hpp — E4..E7: the forward of ONE question (at most 32 packed tokens) as five launches per layer.
//
// The reference embeds one question in front of every search (`EmbeddingProvider.embed_query` -> `embed([q])`, reference
// src/utils/embedding_provider.py:118-157; called at src/rag/retriever.py:150-154, 212, 377): 24 layers of XLM-R-large over ~20
// tokens. The arithmetic is nothing (0.6 GFLOP); the work is reading 604 MB of fp16 weights once and ~120 dependent hand-offs of a
// [tokens][hidden] activation block between the stages. Round 3 ran that as ~170 launches (7 per layer: three projections through
// k_enc_linear_small on 64 - 256 workgroups, attention, two add + LayerNorm kernels). Here a layer is FIVE launches, every
// projection wide enough for the whole chip, and no stand-alone LayerNorm / add kernels:
//
//   E4 k_enc_stage<LNPRO>   qkv = LN(s) Wqkv^T + b        s: the previous block's pre-LayerNorm sum (fp16), LayerNorm in the
//                                                         prologue of EVERY workgroup (its weight fragments are already in flight:
//                                                         they depend on nothing); workgroup r also stores row r of y = LN(s)
//   E5 k_enc_attn_small     ctx = softmax(q k^T) v        one workgroup per head, QK^T and PV on v_mfma_f32_16x16x32_f16
//   E4 k_enc_stage<EPI 2>   s1 = y + (ctx Wo^T + b)       residual add in the epilogue (fp16, as the module's add does)
//   E4 k_enc_stage<LNPRO,1> f  = gelu(LN(s1) W1^T + b)    y1 = LN(s1) stored the same way
//   E4 k_enc_stage<EPI 2>   s2 = y1 + (f W2^T + b)
//
// plus E6 k_enc_embed (the three embedding rows of a token summed) in front and E7 k_enc_ln_rows (the last LayerNorm, CLS rows
// only, widened to fp32) behind. Every stage kernel: one workgroup of 16 waves per FPB output features (16, or 8 / 4 of the MFMA
// tile's 16 rows so that a 1024-feature projection still fills 128 / 256 CUs), the waves split K, each streams its slice of the
// weight rows straight into MFMA A fragments with everything in flight before the first MFMA, activations are the B operand (token
// on the lane), the 16 partial tiles meet in LDS slabs and are added in wave order (deterministic), bias / GELU / residual in fp32.
// Roofline: HBM, N*K*2 B per launch (the weight matrix once); what the kernel actually waits for is latency.
// (Measured and dropped, DESIGN.md §10: the idle waves touching the NEXT stage's weights — every kernel then ends later by those
// loads' latency, +0.08 ms per question — and the same chain as kernels on two streams handing over through counters in device memory,
// 1.46 against 0.95 ms: profiles/r04/enc_chain_*.)
#pragma once
#include "enc_kernels.hpp"

namespace rdx {

struct EncStage {
    const _Float16* x;       // LNPRO: s [T][K], the pre-LayerNorm sums. Otherwise activations [rows][K]
    const int64_t* x_rows;   // plain input only, or NULL: stage token t reads row x_rows[t] of x and of res (the last layer's CLS rows)
    const _Float16* gamma;   // LNPRO: LayerNorm weight, bias [K]
    const _Float16* beta;
    float eps;
    _Float16* y_out;         // LNPRO: y = LayerNorm(s) [T][K] (row r written by workgroup r % grid), or NULL
    const _Float16* w;       // [N][K]
    const _Float16* bias;    // [N]
    const _Float16* res;     // EPI 2: residual [rows][N]
    _Float16* out;           // [T][N]
    int T, N;
};

constexpr int ENC_EPI_BIAS = 0, ENC_EPI_GELU = 1, ENC_EPI_RESIDUAL = 2;

__device__ __forceinline__ float enc_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// LayerNorm of one row held as NCH chunks of 8 halves per lane (element c*512 + lane*8 + e): fp32 statistics, biased variance —
// the arithmetic of k_enc_add_ln behind its add
template <int NCH>
__device__ __forceinline__ void enc_ln_row(const h8 (&v)[NCH], const _Float16* gamma, const _Float16* beta, float eps, int lane, h8 (&y)[NCH]) {
    constexpr int HID = NCH * 512;
    float x[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[c][e] = (float)v[c][e];
            sum += x[c][e];
        }
    const float mean = enc_wave_sum(sum) * (1.f / HID);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = x[c][e] - mean;
            sq += d * d;
        }
    const float rstd = rsqrtf(enc_wave_sum(sq) * (1.f / HID) + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cc = c * 512 + lane * 8;
        const h8 gw = *reinterpret_cast<const h8*>(gamma + cc), bw = *reinterpret_cast<const h8*>(beta + cc);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[c][e] = (_Float16)((x[c][e] - mean) * rstd * (float)gw[e] + (float)bw[e]);
    }
}

// NTB 16-token blocks (1, 2); K = NPH phases of KC = KCS * 512 input features (512 | 1024 | 2 x 1024 | 4 x 1024); FPB output features
// per workgroup. grid N / FPB, 1024 threads = 16 waves; in every phase wave w multiplies k's [64 w .. 64 w + 63] (KCS = 2) of the phase.
// The activation block of a phase, [NTB*16][KC], is read ONCE per workgroup in whole rows (128-byte lines; round 4's first version took
// its B fragments straight from global memory as 16 rows x 64 bytes per instruction: twice the work for the CU's load path, 13 us for
// FFN-down's 256 KB) into an LDS image — LayerNorm'd on the way for LNPRO — while the next phase's rows are already on their way to
// registers; 16-byte chunk ci of row r sits in slot ci ^ (r & 15) (a fragment read takes the same chunk of 16 rows: 16 bank slots).
// Dynamic LDS: slabs NTB * 16 KiB | image NTB * 16 * KC * 2 bytes.
template <int NTB, int KCS, int NPH, int FPB, bool LNPRO, int EPI>
__global__ __launch_bounds__(1024) void k_enc_stage(const EncStage a) {
    extern __shared__ __attribute__((aligned(16))) char st_smem[];
    static_assert(!LNPRO || NPH == 1, "the LayerNorm prologue needs whole rows in one phase");
    constexpr int KC = KCS * 512, K = KC * NPH, KQ = KCS * 32;
    constexpr int SLAB_BYTES = NTB * 16384;
    float* slab = reinterpret_cast<float*>(st_smem);   // [16 waves][NTB][4][64]
    _Float16* img = reinterpret_cast<_Float16*>(st_smem + SLAB_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * FPB;
    const int T = a.T;

    // the weights first: they depend on nothing this launch or the one before it computes. All of them in flight at once.
    const _Float16* wp = a.w + (int64_t)(n0 + (l15 & (FPB - 1))) * K + wave * KQ + lq * 8;
    half8 A[NPH][KCS];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int s = 0; s < KCS; ++s) A[ph][s] = *reinterpret_cast<const half8*>(wp + ph * KC + s * 32);

    // this wave's rows of the activation block: wave + 16 i
    const _Float16* xr[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) {
        const int r = wave + i * 16;
        const int64_t row = (!LNPRO && a.x_rows && r < T) ? a.x_rows[r] : (int64_t)r;
        xr[i] = a.x + row * K + lane * 8;
    }
    h8 raw[NTB][KCS];
    auto load_phase = [&](int ph) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTB; ++i)
            if (wave + i * 16 < T) {
#pragma unroll
                for (int c = 0; c < KCS; ++c) raw[i][c] = *reinterpret_cast<const h8*>(xr[i] + ph * KC + c * 512);
            }
    };
    load_phase(0);
    // what the epilogue's threads will want (waves 0..3: accumulator register r = wave of every lane), requested now
    const int f_loc = lq * 4 + wave;                        // feature inside the MFMA tile's 16 rows
    const bool f_ok = wave < 4 && f_loc < FPB;
    float bv = 0.f;
    _Float16 rv[NTB];
    if (f_ok) {
        bv = (float)a.bias[n0 + f_loc];
        if constexpr (EPI == ENC_EPI_RESIDUAL) {
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) {
                const int t = tb * 16 + l15;
                const int tt = t < T ? t : T - 1;
                const int64_t row = a.x_rows ? a.x_rows[tt] : (int64_t)tt;
                rv[tb] = a.res[row * a.N + n0 + f_loc];
            }
        }
    }

    f32x4 acc[NTB];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
#pragma unroll
        for (int i = 0; i < NTB; ++i) {
            const int r = wave + i * 16;
            h8 y[KCS];
            if (r < T) {
                if constexpr (LNPRO) {
                    enc_ln_row<KCS>(raw[i], a.gamma, a.beta, a.eps, lane, y);
                    if (a.y_out && (r % (int)gridDim.x) == (int)blockIdx.x) {
#pragma unroll
                        for (int c = 0; c < KCS; ++c) *reinterpret_cast<h8*>(a.y_out + (int64_t)r * K + c * 512 + lane * 8) = y[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < KCS; ++c) y[c] = raw[i][c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < KCS; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[c][e] = (_Float16)0.f;
            }
#pragma unroll
            for (int c = 0; c < KCS; ++c) {
                const int ci = c * 64 + lane;
                *reinterpret_cast<h8*>(img + r * KC + ((ci ^ (r & 15)) * 8)) = y[c];
            }
        }
        __syncthreads();
        if (ph + 1 < NPH) load_phase(ph + 1);      // the next phase's rows travel while this one is multiplied
        half8 B[NTB][KCS];
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
            for (int s = 0; s < KCS; ++s) {
                const int ci = wave * (KQ / 8) + s * 4 + lq;
                B[tb][s] = *reinterpret_cast<const half8*>(img + (tb * 16 + l15) * KC + ((ci ^ l15) * 8));
            }
#pragma unroll
        for (int s = 0; s < KCS; ++s)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ph][s], B[tb][s], acc[tb], 0, 0, 0);
        if (ph + 1 < NPH) __syncthreads();          // every fragment of this phase is in registers: the image may be overwritten
    }

    // D[feature = lq * 4 + r][token = l15]: park the partial tile; waves 0..3 add the sixteen in wave order
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[((wave * NTB + tb) * 4 + r) * 64 + lane] = acc[tb][r];
    __syncthreads();
    if (!f_ok) return;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const int t = tb * 16 + l15;
        float v = bv;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) v += slab[((wv * NTB + tb) * 4 + wave) * 64 + lane];
        _Float16 o;
        if constexpr (EPI == ENC_EPI_GELU) {
            o = (_Float16)(0.5f * v * (1.f + erff(v * 0.70710678118654752f)));
        } else if constexpr (EPI == ENC_EPI_RESIDUAL) {
            o = (_Float16)((float)(_Float16)v + (float)rv[tb]);   // the projection rounded to fp16, then an fp16 add: what the module computes
        } else {
            o = (_Float16)v;
        }
        if (t < T) a.out[(int64_t)t * a.N + n0 + f_loc] = o;
    }
}

// E5: attention of at most 32 packed tokens, one workgroup per head, wave qt owns queries 16 qt .. 16 qt + 15.
//   S^T = K Q^T  (A = keys x dims, B = dims x queries: both fragments are 16 contiguous bytes of a qkv row) -> the lane holds, for ITS
//   query (column), the scores of keys 4g + r and 16 + 4g + r (g = lane >> 4): the soft-max of a query is 8 local values and two
//   cross-lane steps; masked (another text's key, or beyond T) = weight 0. P^T stays in registers and IS the B operand of
//   O^T = V^T P^T (k position 8g + j <-> key 4g + j, 16 + 4g + j - 4: V^T is read from LDS in that order), so P never moves.
//   V^T [dim][key] is staged through LDS with transposing 2-byte writes (2 K elements per head).
// tok_first[t] = index of the first token of t's text: tokens attend to the tokens with the same value.
constexpr int ENC_VT_STRIDE = 36;   // halves per V^T row (32 keys + 4: the 8-byte fragment reads of 16 dims spread over the banks)
template <int NQT>
__global__ __launch_bounds__(64 * NQT) void k_enc_attn_small(const _Float16* __restrict__ qkv, const int32_t* __restrict__ tok_first, int T,
                                                             int heads, float scale_log2, _Float16* __restrict__ ctx) {
    __shared__ __attribute__((aligned(16))) _Float16 vt[64 * ENC_VT_STRIDE];
    __shared__ int tf_s[32];
    const int lane = threadIdx.x & 63, qt = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x;
    const int64_t H = (int64_t)heads * 64, row = 3 * H;
    // fragments of K (A operand) and Q (B operand): requested first
    half8 ak[2][2], bq[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kt * 16 + l15;
        const _Float16* kp = qkv + (int64_t)(key < T ? key : T - 1) * row + H + h * 64 + g * 8;
        ak[kt][0] = *reinterpret_cast<const half8*>(kp);
        ak[kt][1] = *reinterpret_cast<const half8*>(kp + 32);
    }
    const int q = qt * 16 + l15;
    {
        const _Float16* qp = qkv + (int64_t)(q < T ? q : T - 1) * row + h * 64 + g * 8;
        bq[0] = *reinterpret_cast<const half8*>(qp);
        bq[1] = *reinterpret_cast<const half8*>(qp + 32);
    }
    for (int p = threadIdx.x; p < 256; p += 64 * NQT) {   // 16-byte pieces of V: key p >> 3, dims (p & 7) * 8 ..
        const int key = p >> 3, d0 = (p & 7) * 8;
        half8 v;
        if (key < T) v = *reinterpret_cast<const half8*>(qkv + (int64_t)key * row + 2 * H + h * 64 + d0);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;   // (a masked key has weight 0: its value must not be Inf / NaN)
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(d0 + e) * ENC_VT_STRIDE + key] = v[e];
    }
    if (threadIdx.x < 32) tf_s[threadIdx.x] = (int)threadIdx.x < T ? tok_first[threadIdx.x] : -1 - (int)threadIdx.x;
    f32x4 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak[kt][0], bq[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak[kt][1], bq[1], st[kt], 0, 0, 0);
    }
    __syncthreads();
    const int my_tf = tf_s[q];
    float sc[2][4];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = tf_s[kt * 16 + 4 * g + r] == my_tf;
            sc[kt][r] = ok ? st[kt][r] * scale_log2 : -INFINITY;
            m = fmaxf(m, sc[kt][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));   // finite: a token's own key is never masked
    float l = 0.f;
    half8 pb;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(sc[kt][r] - m);
            l += p;
            pb[kt * 4 + r] = (_Float16)p;
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const _Float16* vp = vt + (dt * 16 + l15) * ENC_VT_STRIDE + 4 * g;
        const half4 v0 = *reinterpret_cast<const half4*>(vp), v1 = *reinterpret_cast<const half4*>(vp + 16);
        const half8 av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // O^T[dim 16 dt + 4g + r][query]
        if (q < T) {
            half4 r4;
#pragma unroll
            for (int r = 0; r < 4; ++r) r4[r] = (_Float16)(o[r] * inv);
            *reinterpret_cast<half4*>(ctx + (int64_t)q * H + h * 64 + dt * 16 + 4 * g) = r4;
        }
    }
}

// E6: s0[t] = (word[tok[t]] + pos[pos_id[t]]) + type0, two fp16 adds (the order and the roundings of the module's embedding sum);
// the embedding LayerNorm is the first layer's LNPRO prologue. One wave per token.
template <int NCH>
__global__ __launch_bounds__(256) void k_enc_embed(const int64_t* __restrict__ tok, const int64_t* __restrict__ pos_id,
                                                   const _Float16* __restrict__ word, const _Float16* __restrict__ pos,
                                                   const _Float16* __restrict__ typ, int T, _Float16* __restrict__ out) {
    constexpr int HID = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int64_t wi = tok[t], pi = pos_id[t];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cc = c * 512 + lane * 8;
        const h8 w = *reinterpret_cast<const h8*>(word + wi * HID + cc), p = *reinterpret_cast<const h8*>(pos + pi * HID + cc);
        const h8 ty = *reinterpret_cast<const h8*>(typ + cc);
        h8 s = w + p;
        s = s + ty;
        *reinterpret_cast<h8*>(out + (int64_t)t * HID + cc) = s;
    }
}

// E7: out[r] = (float)(fp16) LayerNorm(s[r]) — the last block's LayerNorm on the CLS rows, widened for K1. One wave per row.
template <int NCH>
__global__ __launch_bounds__(256) void k_enc_ln_rows(const _Float16* __restrict__ s, const _Float16* __restrict__ gamma,
                                                     const _Float16* __restrict__ beta, float eps, int rows, float* __restrict__ out) {
    constexpr int HID = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    h8 v[NCH], y[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = *reinterpret_cast<const h8*>(s + (int64_t)r * HID + c * 512 + lane * 8);
    enc_ln_row<NCH>(v, gamma, beta, eps, lane, y);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float* op = out + (int64_t)r * HID + c * 512 + lane * 8;
        *reinterpret_cast<float4*>(op) = make_float4((float)y[c][0], (float)y[c][1], (float)y[c][2], (float)y[c][3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4((float)y[c][4], (float)y[c][5], (float)y[c][6], (float)y[c][7]);
    }
}

}  // namespace rdx
