// scan_kernel.hpp — K2+K3: the corpus scan. fp16 MFMA GEMM  S[row][query] = <corpus row, query>  over the
// fragment-ordered scan copy, with the top-k work fused into the epilogue so the B x N score matrix never exists in HBM.
//
// Roofline: reads rows*dim_pad*2 B of corpus exactly once per launch (HBM) and does 2*rows*nq*dim_pad flop
// (MFMA). HBM-bound while nq per sweep is small, MFMA-bound from nq ~ 300 up.
//
// Dataflow (one workgroup = 8 waves = one 256-row corpus tile x BN queries per step, persistent over a stream of tiles):
//   corpus  HBM --global_load_dwordx4 (1 KiB per wave-instruction, fragment order)--> VGPRs of the ONE wave that owns those
//           32 rows. No LDS, no sharing between waves, prefetched two k-steps ahead into the registers the matrix pipe
//           has just finished reading. Launches with ONE query tile read every corpus byte once: non-temporal loads.
//   queries L2 --LDS-DMA (global_load_lds_dwordx4)--> 4-slot LDS ring of (BN x 64 k) images shared by the 8 waves, filled
//           three k-steps ahead (waves 0..3 right behind the barrier, waves 4..7 a quarter step later: the two waves of
//           a SIMD never sit in the expensive DMA issue together); when all k-steps of a 64-query tile fit (<= 128 KiB)
//           the tile stays RESIDENT in LDS and the main loop has no DMA and no barrier at all (the HBM-bound regime).
//   MFMA    v_mfma_f32_16x16x32_f16 (same cycles per flop as 32x32x16, but the chip holds a higher clock on it), corpus
//           fragment = A operand, query fragment = B operand: D[row][query] has the QUERY on the lane (col = lane & 15)
//           and 4 corpus rows in the 4 accumulator registers of a block, so the per-query threshold is one value per
//           lane and block and the epilogue is compare-only.
//   The nqt query tiles of one stream run on workgroups with equal blockIdx % 8, i.e. on one XCD, so a corpus tile is
//   fetched over the fabric once or twice and otherwise re-read from that XCD's L2 (speed only — nothing depends on
//   the placement; the optional sibling lock-step below tightens it).
//
// Synchronisation per k-step: ONE s_waitcnt vmcnt(V) at the top (V = vector-memory operations issued during the previous
// step, which stay in flight) and, in ring mode, ONE mid-step s_waitcnt + s_barrier that never waits for memory in
// steady state: everything a step needs was issued two or three steps earlier.
// The corpus loads are inline asm (the compiler would drain vmcnt(0) at their first use next to LDS-DMA,
// cdna_hip_programming.md §5 trap (b)); their destination registers are handed to the compiler only through the
// "+v" operands of the wait statement (§5.7 form (ii)). Consequences the code has to respect: such loads are issued
// unconditionally (a conditional one makes the compiler copy a register that has not landed), both fragment sets stay
// alive until the final vmcnt(0), and the kernel must not spill (rag_dpo_amd/build.py refuses a build that does).
//
// Epilogues.
//   EPI_SETMAX (threshold bootstrap, run on every sample_div-th tile): accumulator blocks keep a running max over all
//     tiles of the stream -> 4 disjoint row sets per (stream, wave) and query, 32 per stream. k_tau takes the k-th
//     largest set max: k DISTINCT rows score at least that, which makes tau = that - 2E a lower bound for every true
//     top-k row's coarse score (E = |coarse - exact| bound).
//   EPI_EMIT (main pass): every (row, query) whose coarse score >= tau[query] is appended to the (query, stream) candidate
//     segment: slot from an LDS counter (no global atomics, no round trip), 8-byte fire-and-forget store.
#pragma once
#include <type_traits>

#include "rdx_common.hpp"

namespace rdx {

#if defined(RDX_ABL_NOA) || defined(RDX_ABL_NOB) || defined(RDX_ABL_NOEMIT) || defined(RDX_ABL_HALFB) || defined(RDX_ABL_DBLA)
#define RDX_EMIT_ON false   // ablation builds never emit (their scores are meaningless)
#else
#define RDX_EMIT_ON true
#endif
// Developer ablations that price a 4 x 2 wave layout (each wave 64 rows x 128 queries) WITHOUT building it — timing only, scores are
// garbage (tools/ab_lib.py). At BN = 256, ring mode:
//   RDX_ABL_HALFB  half the query-fragment reads: odd MFMA groups reuse the even group's fragments (16 instead of 32 ds_read_b128 per
//                  wave and k-step, what a wave with 128 queries would read);
//   RDX_ABL_DBLA   twice the corpus loads: waves w and w + 4 both fetch the two 32-row blocks 2(w & 3), 2(w & 3) + 1 (8 instead of 4
//                  global_load_dwordx4 per wave and k-step, every line requested by two waves of the CU; the second block lands in the
//                  first one's registers — a real build needs 32 more).
#ifdef RDX_ABL_HALFB
#define RDX_HALFB 1
#else
#define RDX_HALFB 0
#endif
#ifdef RDX_ABL_DBLA
#define RDX_DBLA 1
#else
#define RDX_DBLA 0
#endif

#ifndef RDX_PD256
#define RDX_PD256 1   // query-fragment groups read ahead at BN = 256 (each costs 8 VGPRs)
#endif
#ifndef RDX_HALF_STAGGER
#define RDX_HALF_STAGGER 0   // 1: waves 4-7 run half a k-step behind waves 0-3 (their SIMD partners), see `step`; measured -3..-5 % at B = 1024
#endif
#ifndef RDX_DMA_STAGGER
#define RDX_DMA_STAGGER 4   // wave-number mask: waves with (wave & mask) != 0 issue their query-image DMA later in the step
#endif
#ifndef RDX_DMA_LATE_NUM
#define RDX_DMA_LATE_NUM 2  // late position = barrier group + NUM*NG/8 (a quarter step), for the variants with the stand-alone emit check
#endif
#ifndef RDX_DMA_LATE_NUM_FUSED
#define RDX_DMA_LATE_NUM_FUSED 1  // ... and for the fused-check variants: an eighth of a step is +0.2 ... +1 % on every shape tried (c4, c3, a 1.25 M-row
                                  // shard, B = 512), three eighths -6 %. (The stand-alone variants spill two to six registers with 1: they keep 2.)
#endif

#ifndef RDX_BAR_NUM
#define RDX_BAR_NUM 4   // the per-step barrier sits in front of MFMA group RDX_BAR_NUM * NG / 8 (4 = mid-step). Measured at B = 1024, 10 M rows,
                        // same box (barrier group / late-DMA group of 16): 8/12 (this) 1335-1343 TFLOP/s, 2/6 1291-1295, 0/4 1285, 4/6 1275-1280, 10/14 1250-1260
#endif
#ifndef RDX_PRIO
#define RDX_PRIO 2   // which half of the workgroup runs at s_setprio 1: 0 none, 1 waves 4-7, 2 waves 0-3
#endif
#ifndef RDX_ZERO_C
#define RDX_ZERO_C 0   // 1: the first MFMA of a tile takes C = 0 instead of a zeroed accumulator (un-tied destination: the register allocator then spills)
#endif

constexpr int EPI_SETMAX = 0;
constexpr int EPI_EMIT = 1;
constexpr int SETS_PER_STREAM = 32;   // bootstrap sets per (stream, query): 8 waves x 2 lane halves x 2 register classes
constexpr int RING_SLOTS = 4;     // LDS ring of query images: step s, s+1 (certified), s+2, s+3 (in flight)

struct ScanParams {
    const _Float16* shadow;    // fragment-ordered corpus scan copy
    const _Float16* qshadow;   // tiled query scan copy
    int ksteps;                // dim_pad / 64
    int64_t rows;              // valid corpus rows
    int64_t n_tiles;           // ceil(rows / 256)
    int tile_stride;           // 1 (main) or sample_div (bootstrap): tiles 0, stride, 2*stride, ...
    // EPI_SETMAX, option "spread_boot": the sample is every S-th 32-ROW BLOCK instead of every S-th 256-row tile — wave w of schedule
    // entry j scans block (8 j + w) * S = the tile formula's block j * S * 8 + w PLUS w * (S - 1): one per-wave constant on the scan
    // copy's base (wave_off = (S - 1) * bytes of a block) and on the row number (row_off = (S - 1) * 32); span = rows from a tile's
    // first row to the end of its last wave's block (256, or (7 S + 1) * 32). The same number of sampled rows, eight times finer: a
    // run of similar rows stored together (a document's chunks) that fell between two sampled tiles is met by several sampled
    // blocks. The host bounds n_tiles so that every block lies inside the corpus. Tile sample: 0, 0, 256.
    int64_t wave_off;
    int row_off, span;
    int nqt;                   // query tiles of BN queries
    int nq_pad;                // queries padded to 256
    const uint32_t* allow;     // NULL or row bitmap
    // EPI_SETMAX
    float* setmax;             // [nq_pad][n_sets], n_sets = n_streams * SETS_PER_STREAM
    int n_sets;
    // EPI_EMIT
    const float* tau;          // [nq_pad] in accumulator units (score * 4^scale_log2)
    uint32_t* cntw;            // [nq_pad][n_streams] hits of (query, stream); a value above capw means overflow
    uint2* cand;               // [nq_pad][n_streams][capw] (score bits, row)
    uint32_t capw;
    float inv_scale2;          // accumulator -> score
    uint32_t* sib;             // [n_streams][4] progress bytes of the query-tile workgroups of a stream (zeroed per launch), or NULL
    uint8_t* sib_scratch;      // [n_streams][16][8] bytes nobody reads (keeps the per-wave operation counts uniform)
    int use_xlo, bulk_it;      // main pass: after bulk_it interleaved iterations per stream, XCD x owns the schedule range
    int xlo[9];                //   [xlo[x], xlo[x+1]) (the XCDs of one chip do not run equally fast; the host sizes the ranges from the
    unsigned long long* wgt;   //   [grid][2] start / end wall_clock64 of every workgroup (NULL: not wanted)   finish times)
    int sib_lag;               // throttle when the slowest sibling looks more than this many k-steps behind (a snapshot is ~2-3 old)
    int64_t shadow_bytes;      // RDX_CHECK_BOUNDS builds: size of the scan copy ...
    int* oob;                  // ... and the flag a corpus read outside it raises (tests/test_gpu_bounds.py)
};

// 16 B per lane straight into VGPRs; completion is the CALLER's business (counted s_waitcnt vmcnt)
#ifndef RDX_NT_SMALL
#define RDX_NT_SMALL 1
#endif
// NT: the corpus stream of a launch with ONE query tile is read exactly once, by one workgroup -> non-temporal loads (they
// do not displace the query images in L2 and skip the allocate; measured at 10M x 1024: B = 64 6.3 -> 6.95 TB/s, B = 1
// 6.3 -> 7.0, B = 128 6.5 -> 6.8). With several query tiles the siblings WANT the tile in L2: plain loads.
// Address = wave-uniform base (SGPR pair) + 32-bit lane offset + immediate: no per-lane 64-bit address registers or adds.
template <bool NT, int IMM>
__device__ __forceinline__ void gload16(half8& dst, const char* base, uint32_t lane_off) {
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(IMM) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(IMM) : "memory");
}

// a wave-uniform pointer the compiler may have parked in VGPRs: back into SGPRs for an "s" asm operand
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_keep(half8 (&a)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}

// FUSEDT (EMIT, BN = 256, even number of k-steps per tile — the host checks): a tile's emit check rides with the first
// k-step of the next tile instead of interrupting the MFMA stream (see `step`).
template <int BN, int EPI, bool HAS_MASK, bool RES, bool SIBT = false, bool NTT = false, bool FUSEDT = false>
__global__ __launch_bounds__(512) void k_scan(const ScanParams p) {
    constexpr bool FUSED = FUSEDT && EPI == EPI_EMIT;
#ifdef RDX_NT_ALL   // developer A/B (tools/ab_lib.py): non-temporal corpus loads also when several query tiles share a corpus tile through L2.
                    // Measured at 10 M x 1024, B = 1024 (round 3, same box, three alternating rounds): 58.9 k against 66.1 k queries/s, fabric
                    // reads per launch 1.74 x — the sibling workgroups stop finding the tile in L2. Plain loads stay.
    constexpr bool NT_A = true;
#else
    constexpr bool NT_A = RDX_NT_SMALL && NTT;   // host: NTT launches have ONE query tile (every corpus byte is read by one workgroup)
#endif
    constexpr int LATE_NUM = FUSED ? RDX_DMA_LATE_NUM_FUSED : RDX_DMA_LATE_NUM;   // where the late half issues its DMA (see `step`)
    constexpr int B_BYTES = BN * BK * 2;  // one k-step image of this workgroup's queries
    constexpr int NPB = BN / 64;          // 1 KiB DMA pieces per wave per query image
    // Sibling lock-step (speed only). The nqt workgroups of a stream read the same corpus tiles; nothing else keeps them
    // together, and once they drift by more than the L2 can hold (8 streams x 16 KiB per k-step per XCD) every one of
    // them fetches its own copy over the fabric. Each workgroup therefore publishes the number of k-steps it has finished
    // (one byte, mod 256) and every wave reads the four bytes of its sibling group once per k-step — one more counted
    // vector-memory operation, consumed two steps later behind the wait that is there anyway — and naps while the slowest
    // sibling is more than sib_lag steps behind. The nap is bounded: a sibling that never shows up (not co-resident)
    // switches the mechanism off for this wave, so every wave reaches its exit whatever the others do.
    // Measured at 10M x 1024, B = 1024 (same box, alternating): fabric traffic per launch 42.0 GB -> 29.7 GB (2.05x -> 1.45x
    // the algorithmic bytes) at sib_lag 6, queries/s -2.6 % (two more vector-memory instructions per wave and k-step ~1.2 %,
    // the rest is running at the pace of the momentarily slowest sibling). The launch is MFMA-bound at a third of the
    // fabric's bandwidth, so this variant (SIBT) is selected only when option "sib_sync" is set.
    constexpr bool SIB = SIBT && EPI == EPI_EMIT && !RES && BN == 256;
    constexpr int SIBN = SIB ? 2 : 0;                  // its vector-memory operations per wave and k-step: one store, one load
    constexpr bool DBLA = RDX_DBLA && BN == 256 && !RES;   // (ablation, see above)
    constexpr bool HALFB = RDX_HALFB && BN == 256 && !RES;
    constexpr int NA = DBLA ? 8 : 4;                   // corpus loads per wave and k-step
    constexpr int V = NA + (RES ? 0 : NPB) + SIBN;     // vector-memory operations a wave issues per k-step
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
    const bool dma_late = (wave & RDX_DMA_STAGGER) != 0;

    const int n_sched = (int)((p.n_tiles + p.tile_stride - 1) / p.tile_stride);   // tiles in this launch
    // schedule entries of this stream: first, first + every, ...   (entry j = corpus tile j * tile_stride)
    int my_tiles = stream < n_sched ? (n_sched - stream + n_streams - 1) / n_streams : 0;
    // With use_xlo the schedule has two parts: the first bulk_it iterations of every stream are dealt as above (entries
    // [0, bulk_it * n_streams), interleaved over all streams: every XCD gets the same), the rest of the schedule is cut
    // into one contiguous range per XCD, sized by the host so that the faster XCDs take more of it.
    int tail_first = 0, bulk_it = my_tiles;
    if (p.use_xlo) {
        const int ls = slot / p.nqt, cnt = p.xlo[xcd + 1] - p.xlo[xcd];   // my stream's index inside the XCD, the XCD's tail share
        bulk_it = p.bulk_it;
        tail_first = p.xlo[xcd] + ls;
        my_tiles = bulk_it + (cnt > ls ? (cnt - ls + G - 1) / G : 0);
    }
    auto sched_of = [&](int it_i) __attribute__((always_inline)) { return it_i < bulk_it ? stream + it_i * n_streams : tail_first + (it_i - bulk_it) * G; };
    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x] = wall_clock64();
    const int KS = p.ksteps;
    const int total = my_tiles * KS;   // k-steps of this workgroup (host keeps tiles*ksteps < 2^31)
    const uint32_t* sib_word = p.sib ? p.sib + stream * 4 + (qt >> 2) : p.cntw;   // (any valid word when the mechanism is off)
    const int sib_n = p.nqt - (qt & ~3) < 4 ? p.nqt - (qt & ~3) : 4;             // workgroups in my sibling group
    bool sib_on = SIB && p.sib != nullptr && p.nqt > 1;
    const uint8_t* sib_pub = (wave == 0 && p.sib) ? reinterpret_cast<const uint8_t*>(sib_word) + (qt & 3)
                                                  : p.sib_scratch + ((stream * 16 + (qt & 15)) * 8 + wave);
    const int zero_off = 0;
    auto sib_lag = [&](uint32_t word, int s_mine) __attribute__((always_inline)) {   // k-steps the slowest sibling of the snapshot is behind s_mine (mod 256)
        int lag = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = (int)(int8_t)(uint8_t)((uint32_t)s_mine - (word >> (8 * j)));
            if (j < sib_n && d > lag) lag = d;
        }
        return lag;
    };

    const int ring_bytes = (RES ? KS : RING_SLOTS) * B_BYTES;
    uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + ring_bytes);   // [BN] hit counters of this (stream, query tile)
    if constexpr (EPI == EPI_EMIT) {
        for (int i = threadIdx.x; i < BN; i += 512) lcnt[i] = 0;
    }

    // query images of this tile: [ks][BN rows][128 B], rows (qt*BN)%256.. of the 256-row block qt*BN/256
    const char* qbase = reinterpret_cast<const char*>(p.qshadow) + ((int64_t)(qt * BN / 256) * KS) * KSTEP_BYTES +
                        (int64_t)((qt * BN) % 256) * 128 + wave * (NPB * 1024) + lane * 16;
    auto issue_b = [&](int ks_i, int slot_i) __attribute__((always_inline)) {   // this wave's pieces of query image ks_i -> LDS slot slot_i
        const char* src = qbase + (int64_t)ks_i * KSTEP_BYTES;
        char* dst = smem + slot_i * B_BYTES + wave * (NPB * 1024);
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    // corpus fragments of (tile iteration it, k-step ks): 4 consecutive 1 KiB chunks of this wave's 32-row block
    int oob_seen = 0;                              // RDX_CHECK_BOUNDS builds only
    const int64_t rb_bytes = (int64_t)KS * 4096;   // bytes of one 32-row block in the scan copy
    const uint32_t lane16 = (uint32_t)lane * 16;
    const char* const shadow_w = reinterpret_cast<const char*>(p.shadow) + (EPI == EPI_SETMAX ? wave * p.wave_off : (int64_t)0);   // (block sample: see ScanParams)
    auto a_src = [&](int it_i, int ks_i) __attribute__((always_inline)) -> const char* {
        const int64_t tile = (int64_t)sched_of(it_i) * p.tile_stride;
#ifdef RDX_CHECK_BOUNDS
        // test build: every corpus fragment address (prefetches of steps that do not exist included) must lie inside the
        // scan copy; an address outside is remembered (reported once, when the workgroup ends) and replaced, so the run ends
        // with an error code instead of a GPU fault. Branch-free: the address stays wave-uniform scalar arithmetic.
        {
            const int64_t off = (tile * 8 + wave) * rb_bytes + (int64_t)ks_i * 4096 + (EPI == EPI_SETMAX ? wave * p.wave_off : (int64_t)0);
            const bool bad = off < 0 || off + 4096 > p.shadow_bytes;
            oob_seen |= bad ? 1 : 0;
            return reinterpret_cast<const char*>(p.shadow) + (bad ? (int64_t)0 : off);
        }
#endif
        return shadow_w + (tile * 8 + (DBLA ? (wave & 3) * 2 : wave)) * rb_bytes + (int64_t)ks_i * 4096;   // wave-uniform
    };

    // v_mfma_f32_16x16x32_f16: the wave's 32 rows are two 16-row blocks m, the queries NB16 blocks of 16; C layout
    // col = lane & 15 (query), row = (lane >> 4) * 4 + reg
    constexpr int NB16 = BN / 16;
    const int l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[2][NB16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB16; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    float runmax[EPI == EPI_SETMAX ? NB16 : 1][1];
    float* tau_s = reinterpret_cast<float*>(lcnt + BN);   // [BN] thresholds of this query tile (LDS: registers are scarce)
    if constexpr (EPI == EPI_SETMAX) {
#pragma unroll
        for (int n = 0; n < (int)(sizeof(runmax) / sizeof(runmax[0])); ++n)
#pragma unroll
            for (int j = 0; j < (int)(sizeof(runmax[0]) / sizeof(float)); ++j) runmax[n][j] = -INFINITY;
    } else {
        // stored [l15][n] so that one ds_read_b128 fetches the thresholds of four consecutive query blocks of a lane
        for (int i = threadIdx.x; i < BN; i += 512) tau_s[(i & 15) * (BN / 16) + (i >> 4)] = p.tau[qt * BN + i];
    }

    // LDS address of this lane's query fragment for k sub-step kk (16-query block n adds n*2048), swizzled:
    // 16x16x32: row l15 of the 16-query block, 16-B chunk 4*kk32 + lq of the 64-k image row
    const int b_sw = ((((qt * BN) % 256) + l15) >> 1) & 7;
    int b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[kk] = l15 * 128 + (((kk * 4 + lq) ^ b_sw) << 4);

    // ---- per-tile epilogue pieces -----------------------------------------------------------------------------------
    // which rows of this wave's 32-row block of schedule entry it_done may be used (ragged last tile, `where` bitmap)
    auto tile_rows = [&](int it_done, int64_t& row_b, uint32_t& okbits, bool& filt) __attribute__((always_inline)) {
        const int64_t tile = (int64_t)sched_of(it_done) * p.tile_stride;
        row_b = tile * TILE_ROWS + wave * (EPI == EPI_SETMAX ? 32 + p.row_off : 32);   // first row of this wave's 32-row block
        const bool ragged = tile * TILE_ROWS + (EPI == EPI_SETMAX ? p.span : TILE_ROWS) > p.rows;   // tile holds padding rows
        okbits = 0xffffffffu;                                  // bit i: row row_b+i may be used
        if (ragged) {
            const int64_t left = p.rows - row_b;
            okbits = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
        }
        if constexpr (HAS_MASK) {
            if (row_b < p.rows) okbits &= p.allow[row_b >> 5];
        }
        filt = HAS_MASK || ragged;
    };
    // EMIT: does query block n (16 queries on the lanes' columns, 32 rows in the 8 accumulator registers) hold a hit?
    // (v_max3_f32 by hand: fmaxf() makes the compiler quiet every input first — `v_max_f32 x, x, x` six times per block — which the
    //  accumulators, finite sums of finite products, do not need: 4 instead of 10 vector instructions per block)
    auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    };
    auto block_max = [&](int n) __attribute__((always_inline)) {
        const float t1 = max3(acc[0][n][0], acc[0][n][1], acc[0][n][2]);
        const float t2 = max3(acc[1][n][0], acc[1][n][1], acc[1][n][2]);
        return max3(max3(t1, t2, acc[0][n][3]), acc[1][n][3], acc[1][n][3]);
    };
    // EMIT, rare path (a handful of blocks per tile): append the hits of block n to the (query, stream) segments. Slot from an
    // LDS counter (inline asm: next to LDS-DMA the compiler would put s_waitcnt vmcnt(0) in front of an LDS atomic and drain
    // the prefetch pipeline on every hit), 8-byte fire-and-forget store.
    uint2* const cand_s = p.cand;
    const uint32_t capw_s = p.capw;
    auto emit_block = [&](int n, float tq, int it_done) __attribute__((always_inline)) {
        int64_t row_b;
        uint32_t okbits;
        bool filt;
        tile_rows(it_done, row_b, okbits, filt);
        // Everything this path needs is derived from two lane values made opaque HERE: otherwise the compiler computes the
        // 16 per-block query indices and the 8 row-bit masks once per kernel as loop invariants — 30 VGPRs the MFMA loop
        // does not have (they ended up in scratch).
        int lc = l15, lr = lq * 4;
        asm volatile("" : "+v"(lc), "+v"(lr));
        const int ql = n * 16 + lc;
        // candidate slot index in 32 bits (the host keeps nq_pad * n_streams * capw below 2^29 for this kernel): the
        // store address is an SGPR base + a 32-bit lane offset, no 64-bit vector arithmetic
        const uint32_t seg0 = ((uint32_t)(qt * BN + ql) * (uint32_t)n_streams + (uint32_t)stream) * capw_s;
        const uint32_t row0 = (uint32_t)row_b + (uint32_t)lr;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[m][n][r];
                if (v >= tq && (!filt || ((okbits >> (m * 16 + r + lr)) & 1u))) {
                    uint32_t pos;
                    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&lcnt[ql]);
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(lds_addr), "v"(1u) : "memory");
                    if (pos < capw_s) cand_s[seg0 + pos] = make_uint2(__float_as_uint(v * p.inv_scale2), row0 + (uint32_t)(m * 16 + r));
                }
            }
    };
    // thresholds of this lane's query column for blocks n0..n0+3 (one ds_read_b128; stored [l15][block])
    auto tau_quad = [&](int n0) __attribute__((always_inline)) {
        int t = l15 * NB16 + n0;
        asm volatile("" : "+v"(t));   // not a loop invariant worth a register
        return *reinterpret_cast<const f32x4*>(tau_s + t);
    };
    // this lane's threshold for block n: ONE address register per step (tb, made opaque there so that it is not a loop
    // invariant held across the whole kernel), the block is the instruction's immediate offset
    auto tau_one = [&](int tb, int n) __attribute__((always_inline)) { return tau_s[tb + n]; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (total > 0) {
        // Corpus fragments in flight: AD register sets, step s uses set s % AD and refills it with step s + AD. Two sets
        // (two k-steps of prefetch) are all the 256-query kernels have registers for; the resident 64-query kernel (no DMA,
        // no barrier, 140 VGPRs) takes four: its launches are HBM-latency-bound — tiny corpora run ONE or two tiles per
        // workgroup, 16 dependent k-steps each (100 k x 1024, B = 64: main scan 42 -> 2x us, bootstrap 19 -> 1x us).
        constexpr int AD = (RES && BN == 64) ? 4 : 2;
        half8 a0[4], a1[4], a2[4], a3[4];   // (a2, a3 unused and eliminated when AD == 2)
        // ---- prologue: query images of steps 0..2 (ring) or the whole tile (resident), corpus fragments of steps 0..AD-1 ----
        if constexpr (RES) {
            for (int ks_i = 0; ks_i < KS; ++ks_i) issue_b(ks_i, ks_i);
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) issue_b(j % KS, j);   // step j reads k-step image j mod KS (one query tile for all corpus tiles)
        }
        // Every step issues its V loads unconditionally (steps that do not exist re-read the stream's first step: valid
        // memory, never consumed) so that the counted waits stay uniform and no conditional copy of an in-flight
        // register is ever needed. An asm load must not be in flight towards a register the compiler considers dead
        // (it would reuse the register and the late write-back would corrupt it): all fragment sets are kept alive
        // until the final s_waitcnt vmcnt(0) below.
        auto first_load = [&](half8 (&af)[4], int j) __attribute__((always_inline)) {   // fragments of step j (or of step 0 when j does not exist)
            const bool have = j < total;
            const char* sj = a_src(have ? j / KS : 0, have ? j % KS : 0);
            gload16<NT_A, 0>(af[0], sj, lane16);
            gload16<NT_A, 1024>(af[1], sj, lane16);
            gload16<NT_A, 2048>(af[2], sj, lane16);
            gload16<NT_A, 3072>(af[3], sj, lane16);
            if constexpr (DBLA) {
                const char* sj2 = sj + rb_bytes;
                gload16<NT_A, 0>(af[0], sj2, lane16);
                gload16<NT_A, 1024>(af[1], sj2, lane16);
                gload16<NT_A, 2048>(af[2], sj2, lane16);
                gload16<NT_A, 3072>(af[3], sj2, lane16);
            }
        };
        first_load(a0, 0);
        first_load(a1, 1);
        if constexpr (AD == 4) {
            first_load(a2, 2);
            first_load(a3, 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the only barrier that waits for memory: every wave's prologue DMA has landed

        int it = 0, ks = 0;                  // step being computed
        int it2 = AD / KS, ks2 = AD % KS;    // step s+AD (corpus fragments prefetched during step s)
        // its address, kept as a running pointer: + 4096 per k-step, recomputed (three 64-bit multiplies) once per tile instead
        // of in every step — the listing showed ~45 scalar instructions of address arithmetic in each half of a k-step
        // (the bootstrap variants, with their 16 running maxima, spill two registers with it and keep the recomputation)
#ifdef RDX_CHECK_BOUNDS
        constexpr bool RUNNING_A = false;   // (every address the test build computes goes through the checking helper and must be a real one)
#else
        constexpr bool RUNNING_A = EPI == EPI_EMIT;
#endif
        const char* a_next = RUNNING_A ? a_src(it2, ks2) : nullptr;
        const char* const a_first = RUNNING_A ? a_src(0, 0) : nullptr;   // what a step beyond the stream's end re-reads (never consumed)
        int slot_c = 0;                // ring slot of step s (= s mod 4)
        int ksb = 3 % KS;              // k-step image that step s+3 reads (issued during step s)

        // The step's 2*2*NB16 MFMAs run as NG groups of GB query blocks. The query fragments of group g+PD are read from LDS
        // right behind the first MFMA of group g into a 4-deep register ring — also ACROSS the step boundary (the last PD
        // groups of step s prefetch the first PD groups of step s+1), so the matrix pipe never drains between steps.
        // Issue order pinned with sched_barrier(0).
        constexpr int GB = 2;                        // query blocks per group
        constexpr int NKK = 2;                       // k sub-steps of 32
        constexpr int GPK = NB16 / GB;               // groups per k sub-step
        constexpr int QB_BYTES = 2048;               // LDS bytes of one 16-query block
        constexpr int NG = NKK * GPK;                // groups per step (4, 8 or 16: a multiple of the register ring)
        // groups read ahead; at BN = 256 the register file decides
        constexpr int PD = BN >= 256 ? RDX_PD256 : (NG / 2 < 3 ? NG / 2 : 3);
        constexpr int NBUF = 4;
        half8 bf[NBUF][GB];
        auto load_group = [&](const char* img, int g, half8 (&dst)[GB]) __attribute__((always_inline)) {
            const int kk = g / GPK, nb0 = (g % GPK) * GB;
#pragma unroll
            for (int j = 0; j < GB; ++j) dst[j] = *reinterpret_cast<const half8*>(img + b_off[kk] + (nb0 + j) * QB_BYTES);
        };
#pragma unroll
        for (int g = 0; g < PD; ++g) load_group(smem, g, bf[g]);   // step 0 reads slot 0 / k-step image 0

        uint32_t poll0 = 0, poll1 = 0;   // sibling snapshots in flight (even / odd steps)
        // One k-step. FUSE (a std::true_type tag; EMIT only): this is the FIRST k-step of a tile and the accumulators still
        // hold the finished tile `it_prev`: its emit check runs block by block right in front of the MFMAs that start the
        // new tile in that block's registers with C = 0 — the check's few vector instructions issue in the shadow of the
        // matrix pipe, nothing is zeroed, and the matrix pipe never waits for an epilogue (before: all eight waves left the
        // MFMA stream together once per tile for ~290 vector instructions; ablation: 7 % of the launch).
        auto step = [&](auto fuse_tag, half8 (&af)[4], uint32_t& poll, int s, int it_prev) __attribute__((always_inline)) {
            constexpr bool FUSE = decltype(fuse_tag)::value;
            // my corpus fragments of this step have landed (issued AD steps ago); the V operations of each of the AD-1
            // steps since stay in flight. No barrier here: the query image of step s was certified by the mid-step barrier of step s-1.
            wait_vmcnt_keep<V*(AD - 1)>(af);
            if constexpr (SIB) {
                asm volatile("" : "+v"(poll));   // the snapshot requested two steps ago has landed with the fragments
                if (sib_on && sib_lag(__builtin_amdgcn_readfirstlane(poll), s) > p.sib_lag) {
                    int naps = 0;
                    while (true) {
                        __builtin_amdgcn_s_sleep(8);
                        const uint32_t w = __hip_atomic_load(sib_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (sib_lag(__builtin_amdgcn_readfirstlane(w), s) <= p.sib_lag - 3) break;
                        if (++naps > 1000) {   // ~0.5 ms: the sibling is not running beside us; stop caring
                            sib_on = false;
                            break;
                        }
                    }
                }
                // publish "this workgroup has finished s k-steps". EVERY wave issues one byte store per step so that the counted
                // waits are the same for all of them (an uncounted store would make wave 0 wait for a corpus load issued
                // half a step ago, every step: -5 %); waves 1..7 write to a scratch byte nobody reads. Plain store: the byte
                // stays in this XCD's L2, where the siblings' sc1 (L1-bypassing) loads find it; inline asm: a volatile C++
                // store becomes flat_store sc0 sc1 + s_waitcnt vmcnt(0).
                if (lane == 0) asm volatile("global_store_byte %0, %1, %2" ::"v"(zero_off), "v"(s), "s"(uniform_ptr(sib_pub)) : "memory");
            }
            const int ksn = ks + 1 == KS ? 0 : ks + 1;
            const char* st = smem + (RES ? ks : slot_c) * B_BYTES;
            const char* stn = smem + (RES ? ksn : ((slot_c + 1) & 3)) * B_BYTES;   // image of step s+1
            const bool more = s + AD < total;   // step s+AD exists; otherwise re-read this stream's first step (never used)
            const char* an = RUNNING_A ? (more ? a_next : a_first) : a_src(more ? it2 : 0, more ? ks2 : 0);
            float tq_cur = 0.f;   // FUSE: this lane's threshold for the next block to check, fetched one block ahead
            int tb = l15 * NB16;
            if constexpr (FUSE) {
                asm volatile("" : "+v"(tb));
                tq_cur = tau_one(tb, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int kk = g / GPK, nb0 = (g % GPK) * GB;
                if constexpr (!RES) {
                    // ONE barrier per step, and it never waits for memory in steady state. My DMA pieces of image s+1 were
                    // issued two steps ago: 4 + V newer operations may stay in flight. After the barrier every wave's pieces of
                    // image s+1 have landed and every wave has left step s-1, whose ring slot is refilled with image s+3.
                    // RDX_HALF_STAGGER: waves 0-3 meet the barrier in the MIDDLE of their step, waves 4-7 (their SIMD partners)
                    // at the START of theirs, i.e. the late half runs half a k-step behind its partner for the whole launch:
                    // while one wave of a SIMD sits in wait / barrier / DMA issue / emit check, the other one is in the middle
                    // of a pure MFMA stretch (MI355X_MICROARCH.md "Two waves per SIMD" item 9). The same counts hold for both
                    // halves: a late wave issued its pieces of image s+1 at the start of step s-2, 4 + V operations ago.
                    constexpr int BAR_G = RDX_HALF_STAGGER ? NG / 2 : RDX_BAR_NUM * NG / 8;   // group in front of which the step's barrier sits
                    // the counted wait below assumes that a wave's DMA issue and the barrier lie on the same side of the kk = 0 refill
                    static_assert(RDX_HALF_STAGGER || !RDX_DMA_STAGGER || ((BAR_G < NG / 2) == (BAR_G + LATE_NUM * NG / 8 < NG / 2)), "barrier / late DMA position");
                    static_assert(RDX_HALF_STAGGER || (BAR_G <= NG - PD && BAR_G + (RDX_DMA_STAGGER ? LATE_NUM * NG / 8 : 0) < NG), "barrier / late DMA position");
                    const bool here = RDX_HALF_STAGGER ? (g == 0 ? dma_late : (g == NG / 2 ? !dma_late : false)) : g == BAR_G;
                    if ((RDX_HALF_STAGGER && (g == 0 || g == NG / 2)) || (!RDX_HALF_STAGGER && g == BAR_G)) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (here) {
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + SIBN + V) : "memory");
                            __builtin_amdgcn_s_barrier();
#if !defined(RDX_ABL_NOB)
                            if (RDX_HALF_STAGGER || !RDX_DMA_STAGGER || !dma_late) issue_b(ksb, (slot_c + 3) & 3);
#endif
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#if !defined(RDX_ABL_NOB)
                    // (without the half-step stagger) the two waves of a SIMD issue their DMA pieces a quarter step apart
                    if (!RDX_HALF_STAGGER && RDX_DMA_STAGGER && g == BAR_G + LATE_NUM * NG / 8) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (dma_late) issue_b(ksb, (slot_c + 3) & 3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                }
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int n = nb0 + j;
                    if constexpr (FUSE) {
                        if (kk == 0) {
                            // the finished tile's emit check for block n, in front of the MFMAs that overwrite its registers
                            const float mx = block_max(n);
                            const float tq = tq_cur;
                            if (n + 1 < NB16) tq_cur = tau_one(tb, n + 1);   // next block's threshold, one block ahead
                            if (!RDX_EMIT_ON) asm volatile("" ::"v"(mx));   // keep the values alive in ablation builds
                            if (RDX_EMIT_ON && __any(mx >= tq)) {
                                emit_block(n, tq, it_prev);
                            }
#if !RDX_ZERO_C
#pragma unroll
                            for (int m = 0; m < 2; ++m)
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // (HALFB: the even group's fragment, made a new value for the compiler — equal operands on equal accumulators would be merged)
                    if (HALFB && (g & 1)) asm volatile("" : "+v"(bf[(g & ~1) % NBUF][j]));
                    if (RDX_ZERO_C && FUSE && kk == 0) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[g % NBUF][j], zero4, 0, 0, 0);
                    else acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2 * kk], bf[(HALFB ? (g & ~1) : g) % NBUF][j], acc[0][n], 0, 0, 0);
                    if (j == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (!HALFB || ((g + PD) & 1) == 0) {
                            if (g + PD < NG) load_group(st, g + PD, bf[(g + PD) % NBUF]);
                            else load_group(stn, g + PD - NG, bf[(g + PD) % NBUF]);   // first groups of step s+1 (after the mid barrier)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (RDX_ZERO_C && FUSE && kk == 0) acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[g % NBUF][j], zero4, 0, 0, 0);
                    else acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2 * kk + 1], bf[(HALFB ? (g & ~1) : g) % NBUF][j], acc[1][n], 0, 0, 0);
                }
                if ((g % GPK) == GPK - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    // the matrix pipe has read this sub-step's fragments: refill them with those of step s+2 (land during the next step)
#if defined(RDX_ABL_NOA)   // developer ablation (tools/ab_lib.py): timing without the corpus stream, results are garbage
                    asm volatile("" ::"v"(an));
#else
                    if (kk == 0) {
                        gload16<NT_A, 0>(af[0], an, lane16);
                        gload16<NT_A, 1024>(af[1], an, lane16);
                        if constexpr (DBLA) {
                            gload16<NT_A, 0>(af[0], an + rb_bytes, lane16);
                            gload16<NT_A, 1024>(af[1], an + rb_bytes, lane16);
                        }
                    } else {
                        gload16<NT_A, 2048>(af[2], an, lane16);
                        gload16<NT_A, 3072>(af[3], an, lane16);
                        if constexpr (DBLA) {
                            gload16<NT_A, 2048>(af[2], an + rb_bytes, lane16);
                            gload16<NT_A, 3072>(af[3], an + rb_bytes, lane16);
                        }
                    }
                    if constexpr (SIB) {
                        if (kk == 0) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(poll) : "v"(zero_off), "s"(uniform_ptr(sib_word)) : "memory");
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // stand-alone epilogue of tile it_done: bootstrap maxima (SETMAX); emit check of the LAST tile of the stream and of
        // every tile when the check cannot ride with the next tile's first step (odd number of k-steps)
        auto epilogue = [&](int it_done) __attribute__((always_inline)) {
            f32x4 tq4 = zero4, tq4n = zero4;
            int64_t row_b = 0;
            uint32_t okbits = 0xffffffffu;
            bool filt = false;
            if constexpr (EPI == EPI_EMIT) tq4n = tau_quad(0);
            else tile_rows(it_done, row_b, okbits, filt);
#pragma unroll
            for (int n = 0; n < NB16; ++n) {
                if constexpr (EPI == EPI_SETMAX) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[m][n][r];
                            if (filt && !((okbits >> (m * 16 + lq * 4 + r)) & 1u)) v = -INFINITY;
                            runmax[n][0] = fmaxf(runmax[n][0], v);
                        }
                } else {
                    if ((n & 3) == 0) {
                        tq4 = tq4n;
                        if (n + 4 < NB16) tq4n = tau_quad(n + 4);
                    }
                    const float mx = block_max(n);
                    const float tq = tq4[n & 3];
                    if (!RDX_EMIT_ON) asm volatile("" ::"v"(mx));   // keep the MFMAs alive in ablation builds
                    if (RDX_EMIT_ON && __any(mx >= tq)) emit_block(n, tq, it_done);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
            }
        };

        auto advance = [&](int) __attribute__((always_inline)) {
            const bool last_k = ks == KS - 1;
            const int it_done = it;
            if (++ks == KS) { ks = 0; ++it; }
            if (++ks2 == KS) {
                ks2 = 0;
                ++it2;
                if constexpr (RUNNING_A) a_next = a_src(it2, 0);   // (beyond the stream's last tile: an address nobody loads from)
            } else {
                if constexpr (RUNNING_A) a_next += 4096;
            }
            slot_c = (slot_c + 1) & 3;
            if (++ksb == KS) ksb = 0;
            if constexpr (!FUSED) {
                if (last_k) epilogue(it_done);
            }
        };

        // steps alternate between the two fragment register sets
        int s = 0;
        // Static priority for one wave of every SIMD pair (MI355X_MICROARCH.md "Two waves per SIMD" item 4), set once, never
        // flipped: waves 0-3 — the ones that refill the ring right behind the barrier. Same box, alternating, B = 1024 on 10 M
        // rows: 1315 vs 1292-1308 TFLOP/s (+0.5 ... +1.6 %); the other half instead (RDX_PRIO 1): +0 ... +0.5 %; c3 unchanged.
        if (RDX_PRIO != 0 && (RDX_PRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
        if constexpr (FUSED) {
            // Tile by tile (KS is even: every tile starts on the a0 register set). No branch ever chooses between two step
            // bodies (the register allocator answers that with a second copy of the accumulators): the first tile is peeled.
            auto pair = [&](auto first_tag, int it_prev) __attribute__((always_inline)) {
                step(first_tag, a0, poll0, s, it_prev);
                advance(s);
                step(std::false_type{}, a1, poll1, s + 1, 0);
                advance(s + 1);
                s += 2;
            };
            pair(std::false_type{}, 0);
            for (int j = 2; j < KS; j += 2) pair(std::false_type{}, 0);
            for (int t = 1; t < my_tiles; ++t) {
                pair(std::true_type{}, t - 1);
                for (int j = 2; j < KS; j += 2) pair(std::false_type{}, 0);
            }
            epilogue(my_tiles - 1);
        } else {
            if constexpr (AD == 4) {
                for (; s + 3 < total; s += 4) {
                    step(std::false_type{}, a0, poll0, s, 0);
                    advance(s);
                    step(std::false_type{}, a1, poll1, s + 1, 0);
                    advance(s + 1);
                    step(std::false_type{}, a2, poll0, s + 2, 0);
                    advance(s + 2);
                    step(std::false_type{}, a3, poll1, s + 3, 0);
                    advance(s + 3);
                }
                if (s < total) {
                    step(std::false_type{}, a0, poll0, s, 0);
                    advance(s);
                    if (s + 1 < total) {
                        step(std::false_type{}, a1, poll1, s + 1, 0);
                        advance(s + 1);
                        if (s + 2 < total) {
                            step(std::false_type{}, a2, poll0, s + 2, 0);
                            advance(s + 2);
                        }
                    }
                }
            } else {
                for (; s + 1 < total; s += 2) {
                    step(std::false_type{}, a0, poll0, s, 0);
                    advance(s);
                    step(std::false_type{}, a1, poll1, s + 1, 0);
                    advance(s + 1);
                }
                if (s < total) {
                    step(std::false_type{}, a0, poll0, s, 0);
                    advance(s);
                }
            }
        }
        // drain the never-consumed tail prefetches; naming all eight fragments keeps their registers reserved until here
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3])
                     : "memory");
        if constexpr (AD == 4)
            asm volatile("" ::"v"(a2[0]), "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(a3[0]), "v"(a3[1]), "v"(a3[2]), "v"(a3[3]));
        if constexpr (SIB) asm volatile("" ::"v"(poll0), "v"(poll1));
    }

    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x + 1] = wall_clock64();
#ifdef RDX_CHECK_BOUNDS
    if (oob_seen && lane == 0) atomicOr(p.oob, 1);
#endif
    if constexpr (EPI == EPI_SETMAX) {
        // set id = (stream*8 + wave)*4 + (lane >> 4) ; layout setmax[query][set]
#pragma unroll
        for (int n = 0; n < NB16; ++n)
            p.setmax[(int64_t)(qt * BN + n * 16 + l15) * p.n_sets + (int64_t)(stream * 8 + wave) * 4 + lq] = runmax[n][0];
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < BN; i += 512) p.cntw[(int64_t)(qt * BN + i) * n_streams + stream] = lcnt[i];
    }
}

// K2b. Threshold bootstrap for SMALL launches (<= 64 queries, at most a few 32-row blocks per CU): the K loop split over the waves.
// The scan kernel above samples whole 256-row tiles: a launch that samples 33 tiles (100 k rows, the sample is 8 K rows) keeps
// 33 of 256 CUs busy with 16 DEPENDENT k-steps each — 19 us of latency for 0.3 us of arithmetic (round 2's c2: the bootstrap
// was a fifth of the step). Here one workgroup takes ONE 32-row block of the scan copy, wave w takes the k-steps w, w+8, ...
// of it (all their fragment loads in flight at once: no dependent chain), the eight partial 32 x 64 tiles are summed in LDS
// (ds_add_f32), and the block's four 8-row sets leave their per-query maxima: 256 workgroups x 32 rows sample the same 8 K
// rows in one round of two k-steps. The query fragments come straight from the tiled query scan copy into registers (each
// fragment is used once; no LDS image). The partial sums are added in another order than the main scan adds them: both are
// within E of the exact score (E bounds fp32 accumulation in ANY order), which is all the threshold's proof uses.
//   unit u of U -> 32-row block rb = floor(u * n_blocks32 / U); set id = u * 4 + j, j = 8-row group; setmax[query][n_sets]
struct BootParams {
    const _Float16* shadow;
    const _Float16* qshadow;
    int ksteps;
    int64_t rows;
    int64_t n_blocks32;
    int units;
    const uint32_t* allow;
    float* setmax;
    int n_sets;
};
constexpr int BOOT_BN = 64;
constexpr int BOOT_LD = 32 + 4;   // floats per query in a partial-sum slab [query][row]: a lane's 4 rows are one ds_write_b128, and the 16 lanes of
                                  // a pass land on 16 distinct bank quads (9 * query mod 16)
__global__ __launch_bounds__(512) void k_boot(const BootParams p) {
    // The eight waves' partial tiles meet in LDS WITHOUT atomics: ds_add_f32 measured ~125 cycles per wave-instruction on this chip
    // (256 of them per workgroup: 20 of the first version's 27 us). Waves 4-7 park their tiles, waves 0-3 add their partner's to
    // their own and park the sums, and the final pass adds the four slabs.
    __shared__ __attribute__((aligned(16))) float slab[4][BOOT_BN * BOOT_LD];
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t rb = (int64_t)u * p.n_blocks32 / p.units;
    const int KS = p.ksteps;
    constexpr int NB = BOOT_BN / 16;
    f32x4 acc[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* a_base = reinterpret_cast<const char*>(p.shadow) + rb * ((int64_t)KS * 4096) + lane * 16;
    const char* q_base = reinterpret_cast<const char*>(p.qshadow);   // query block 0 (<= 64 queries): image row r, 16-B chunk c -> slot c ^ ((r >> 1) & 7)
    // two k-steps per round: 8 corpus + 16 query fragments (96 VGPRs) requested together, then 32 MFMAs
    for (int ks0 = wave; ks0 < KS; ks0 += 16) {
        half8 a[2][4], b[2][2][NB];
        const int ks1 = ks0 + 8 < KS ? ks0 + 8 : ks0;   // (no second step: re-read the first, its products are dropped below)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ks = t ? ks1 : ks0;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][c] = *reinterpret_cast<const half8*>(a_base + (int64_t)ks * 4096 + c * 1024);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int r = n * 16 + l15;
                    b[t][kk][n] = *reinterpret_cast<const half8*>(q_base + ((int64_t)ks * 256 + r) * 128 + (((kk * 4 + lq) ^ ((r >> 1) & 7)) << 4));
                }
        }
        // all 24 requests are out before the first MFMA asks for one (left to itself the scheduler interleaves load, wait, MFMA
        // with two loads in flight: 24 dependent round trips, the very chain this kernel exists to avoid)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && ks0 + 8 >= KS) break;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * kk], b[t][kk][n], acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * kk + 1], b[t][kk][n], acc[1][n], 0, 0, 0);
                }
        }
    }
    // acc[m][n] = rows m*16 + lq*4 + 0..3 of query n*16 + l15  ->  slab[query][row]
    float* mine = slab[wave & 3] + l15 * BOOT_LD + lq * 4;
    if (wave >= 4) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) *reinterpret_cast<f32x4*>(mine + n * 16 * BOOT_LD + m * 16) = acc[m][n];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f32x4* cell = reinterpret_cast<f32x4*>(mine + n * 16 * BOOT_LD + m * 16);
                const f32x4 o = *cell;
                *cell = f32x4{acc[m][n][0] + o[0], acc[m][n][1] + o[1], acc[m][n][2] + o[2], acc[m][n][3] + o[3]};
            }
    }
    __syncthreads();
    // which of the block's 32 rows count (ragged end of the corpus, `where` bitmap: one word per 32-row block)
    const int64_t row0 = rb * 32;
    uint32_t ok = 0xffffffffu;
    const int64_t left = p.rows - row0;
    if (left < 32) ok = left <= 0 ? 0u : ((1u << left) - 1u);
    if (p.allow && left > 0) ok &= p.allow[rb];
    if (threadIdx.x < 4 * BOOT_BN) {
        const int q = threadIdx.x & (BOOT_BN - 1), j = threadIdx.x / BOOT_BN;   // set j = rows 8j .. 8j+7 of the block
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(slab[w] + q * BOOT_LD + j * 8);
            const f32x4 y = *reinterpret_cast<const f32x4*>(slab[w] + q * BOOT_LD + j * 8 + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] += x[r];
                hi[r] += y[r];
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mx = fmaxf(mx, ((ok >> (j * 8 + r)) & 1u) ? lo[r] : -INFINITY);
            mx = fmaxf(mx, ((ok >> (j * 8 + 4 + r)) & 1u) ? hi[r] : -INFINITY);
        }
        p.setmax[(int64_t)q * p.n_sets + (int64_t)u * 4 + j] = mx;
    }
}

// K2c. Main scan of SMALL launches (<= 64 queries, <= 16 k-steps per row, at most a few hundred 32-row blocks per CU): k_boot's split-K
// dataflow over ALL rows, with the emit epilogue. Why: the streaming kernel above deals whole 256-row tiles to 256 workgroups —
// 100 k rows are 391 tiles, 1.53 per workgroup, i.e. TWO rounds of a 16-step pipeline for 1.5 rounds of work (and no finer unit
// helps: 6 250 16-row blocks on 2 048 waves are 3.05 each, i.e. 4) — 41.9 us for a 25.6 us HBM stream at c2. Here a workgroup
// takes a contiguous range of 32-row blocks (3 125 blocks / 256 = 12.2: the longest range is 13, 6 % over the mean), wave w owns
// k-steps w and w + 8 of every block, its query fragments stay in registers for the whole launch (64 VGPRs, loaded once), the
// next block's corpus fragments are requested before the current block is multiplied (64 KB in flight per CU), the eight partial
// 32 x 64 tiles are summed through LDS slabs exactly as in k_boot (same order: the same coarse scores, bit for bit) and 256
// threads compare the sums with the thresholds and append the hits (LDS counter per query, 8-byte store).
struct SmallScanParams {
    const _Float16* shadow;
    const _Float16* qshadow;
    int ksteps;
    int64_t rows;
    int64_t n_blocks32;
    const uint32_t* allow;
    const float* tau;          // [>= 64] accumulator units
    uint32_t* cntw;            // [nq_pad][n_streams]
    uint2* cand;               // [nq_pad][n_streams][capw]
    uint32_t capw;
    float inv_scale2;
    unsigned long long* wgt;   // [grid][2] start / end stamps (NULL: not wanted)
};
__global__ __launch_bounds__(512) void k_scan_small(const SmallScanParams p) {
    __shared__ __attribute__((aligned(16))) float slab[4][BOOT_BN * BOOT_LD];
    __shared__ uint32_t lcnt[BOOT_BN];
    __shared__ float tau_s[BOOT_BN];
    const int stream = blockIdx.x, n_streams = gridDim.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x] = wall_clock64();
    if (threadIdx.x < BOOT_BN) {
        lcnt[threadIdx.x] = 0;
        tau_s[threadIdx.x] = p.tau[threadIdx.x];
    }
    const int KS = p.ksteps;   // <= 16 (host)
    constexpr int NB = BOOT_BN / 16;
    const bool has0 = wave < KS, has1 = wave + 8 < KS;
    const int ks0 = has0 ? wave : 0, ks1 = has1 ? wave + 8 : ks0;
    const int64_t b0 = (int64_t)stream * p.n_blocks32 / n_streams, b1 = (int64_t)(stream + 1) * p.n_blocks32 / n_streams;
    // this wave's query fragments, for the whole launch
    half8 b[2][2][NB];
    {
        const char* q_base = reinterpret_cast<const char*>(p.qshadow);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int r = n * 16 + l15;
                    b[t][kk][n] = *reinterpret_cast<const half8*>(q_base + ((int64_t)(t ? ks1 : ks0) * 256 + r) * 128 + (((kk * 4 + lq) ^ ((r >> 1) & 7)) << 4));
                }
    }
    const char* a_lane = reinterpret_cast<const char*>(p.shadow) + lane * 16;
    const int64_t rb_bytes = (int64_t)KS * 4096;
    auto load_a = [&](int64_t rb, half8 (&a)[2][4]) __attribute__((always_inline)) {
        const char* base = a_lane + rb * rb_bytes;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][c] = *reinterpret_cast<const half8*>(base + (int64_t)(t ? ks1 : ks0) * 4096 + c * 1024);
    };
    float* mine = slab[wave & 3] + l15 * BOOT_LD + lq * 4;
    // one block: multiply (fragments `a`), request the next block's fragments into `an` first
    auto block = [&](int64_t rb, half8 (&a)[2][4], half8 (&an)[2][4]) __attribute__((always_inline)) {
        load_a(rb + 1 < b1 ? rb + 1 : rb, an);   // (the last block re-reads itself: no branch around loads)
        __builtin_amdgcn_sched_barrier(0);       // the requests leave before the MFMAs wait for the current fragments
        f32x4 acc[2][NB];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 0 ? !has0 : !has1) continue;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * kk], b[t][kk][n], acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * kk + 1], b[t][kk][n], acc[1][n], 0, 0, 0);
                }
        }
        if (wave >= 4) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) *reinterpret_cast<f32x4*>(mine + n * 16 * BOOT_LD + m * 16) = acc[m][n];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x4* cell = reinterpret_cast<f32x4*>(mine + n * 16 * BOOT_LD + m * 16);
                    const f32x4 o = *cell;
                    *cell = f32x4{acc[m][n][0] + o[0], acc[m][n][1] + o[1], acc[m][n][2] + o[2], acc[m][n][3] + o[3]};
                }
        }
        __syncthreads();
        if (threadIdx.x < 4 * BOOT_BN) {
            const int q = threadIdx.x & (BOOT_BN - 1), j = threadIdx.x / BOOT_BN;   // rows 8j .. 8j+7 of the block, query q
            const int64_t row0 = rb * 32;
            uint32_t ok = 0xffffffffu;
            const int64_t left = p.rows - row0;
            if (left < 32) ok = left <= 0 ? 0u : ((1u << left) - 1u);
            if (p.allow && left > 0) ok &= p.allow[rb];
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(slab[w] + q * BOOT_LD + j * 8);
                const f32x4 y = *reinterpret_cast<const f32x4*>(slab[w] + q * BOOT_LD + j * 8 + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    lo[r] += x[r];
                    hi[r] += y[r];
                }
            }
            const float tq = tau_s[q];
            const uint32_t seg0 = ((uint32_t)q * (uint32_t)n_streams + (uint32_t)stream) * p.capw;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = r < 4 ? lo[r & 3] : hi[r & 3];
                if (v >= tq && ((ok >> (j * 8 + r)) & 1u)) {
                    const uint32_t pos = atomicAdd(&lcnt[q], 1u);
                    if (pos < p.capw) p.cand[seg0 + pos] = make_uint2(__float_as_uint(v * p.inv_scale2), (uint32_t)(row0 + j * 8 + r));
                }
            }
        }
        __syncthreads();   // the slabs are free for the next block
    };
    half8 a0[2][4], a1[2][4];
    if (b0 < b1) load_a(b0, a0);
    __syncthreads();       // lcnt / tau_s initialised
    int64_t rb = b0;
    for (; rb + 1 < b1; rb += 2) {
        block(rb, a0, a1);
        block(rb + 1, a1, a0);
    }
    if (rb < b1) block(rb, a0, a1);
    if (threadIdx.x < BOOT_BN) p.cntw[(int64_t)threadIdx.x * n_streams + stream] = lcnt[threadIdx.x];
    if (p.wgt && threadIdx.x == 0) p.wgt[2 * blockIdx.x + 1] = wall_clock64();
}

// K3a. tau[q] = (k-th largest of the query's set maxima) - 2E, in accumulator units; -inf if fewer than k
// non-empty sets exist (then every allowed row is emitted). One block per query (padding queries: +inf).
// Only the first n_sets_used sets (streams that scanned at least one tile) are looked at.
// k = rank taken: the search's k (provable: k distinct sampled rows reach the value, two_e_scaled = 2E) or a smaller rank with
// two_e_scaled = 0 (speculative threshold, verified by k_refine; rdx_api.hip spec_rank).
__global__ __launch_bounds__(256) void k_tau(const float* __restrict__ setmax, int n_sets, int n_sets_used, int k,
                                             float two_e_scaled, int nq, float* __restrict__ tau) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[HIST_WORDS];
    __shared__ uint32_t bc[4];
    const int q = blockIdx.x;
    if (q >= nq) {   // padding query (zero vector): must never emit
        if (threadIdx.x == 0) tau[q] = INFINITY;
        return;
    }
    const float* sm = setmax + (int64_t)q * n_sets;
    if (k > n_sets_used) {
        if (threadIdx.x == 0) tau[q] = -INFINITY;
        return;
    }
    int64_t n_gt;
    uint32_t key;
    // Every radix pass used to re-read the maxima from global memory, one dependent load per key and thread: they are loaded once,
    // all loads in flight (clamped index, value masked afterwards: a guarded load becomes a branch per load), RN keys per thread
    // (256 threads; 8 for up to 2048 maxima, 32 for up to 8192; more are re-read per pass as before).
    auto in_regs = [&](auto rn_tag) __attribute__((always_inline)) {
        constexpr int RN = decltype(rn_tag)::value;
        float vreg[RN];
        uint32_t kreg[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int i = j * 256 + (int)threadIdx.x;
            vreg[j] = sm[i < n_sets_used ? i : n_sets_used - 1];
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) asm volatile("" : "+v"(vreg[j]));
#pragma unroll
        for (int j = 0; j < RN; ++j) kreg[j] = f2key(vreg[j]);
        return block_kth_largest_scan(
            [&](auto f) {
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    if (j * 256 + (int)threadIdx.x < n_sets_used) f(kreg[j]);
            },
            k, hist, bc, &n_gt);
    };
    if (n_sets_used <= 8 * 256) key = in_regs(std::integral_constant<int, 8>{});
    else if (n_sets_used <= 32 * 256) key = in_regs(std::integral_constant<int, 32>{});
    else key = block_kth_largest([&](int64_t i) { return f2key(sm[i]); }, n_sets_used, k, hist, bc, &n_gt);
    if (threadIdx.x == 0) {
        const float v = key2f(key);
        tau[q] = v > -INFINITY ? v - two_e_scaled : -INFINITY;
    }
}

}  // namespace rdx
