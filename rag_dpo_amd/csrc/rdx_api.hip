// rdx_api.hip — host side of librdx: the C-ABI of include/rdx.h, HBM ownership, kernel dispatch.
// Built for gfx950 only: hipcc --offload-arch=gfx950 -O3 -shared -fPIC rdx_api.hip -o librdx.so
#include "../../include/rdx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <sched.h>
#include <unordered_map>
#include <vector>

#include <immintrin.h>

#include "enc_kernels.hpp"
#include "enc_small.hpp"
#include "k_rows.hpp"
#include "refine_kernel.hpp"
#include "scan_kernel.hpp"
#include "scan_w4.hpp"

using namespace rdx;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            const int _code = (_e == hipErrorOutOfMemory) ? RDX_ERR_NOMEM : RDX_ERR_HIP;           \
            return fail(_code, std::string(#expr) + ": " + hipGetErrorString(_e));                \
        }                                                                                          \
    } while (0)

#define RDX_TRY(expr)              \
    do {                           \
        int _r = (expr);           \
        if (_r != RDX_OK) return _r; \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return RDX_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = need;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            p = nullptr;
            return fail(RDX_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        }
        bytes = want;
        return RDX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }   // error paths that return early do not leak scratch
};

// Host callers: where the results go (see search_chunk)
struct HostOut {
    float* score;
    int64_t* row;
    int32_t* count;
    bool stale;
};

// A search whose kernels (up to k_finish) are enqueued and whose host half — waiting for the mailbox, copying small
// results out, re-weighting the XCD shares, the fallback passes for overflowed queries, the statistics — has not run yet.
// rdx_search runs that half at once; rdx_search_async leaves it to rdx_search_wait, so that the caller can enqueue what
// consumes the results (the RCCL all-gather and the merge) while the scan is still running.
struct PendingSearch {
    bool active = false;
    const float* d_queries = nullptr;
    int64_t nq = 0;
    int k = 0;
    const uint32_t* d_allow = nullptr;
    float* d_score = nullptr;
    int64_t* d_row = nullptr;
    int32_t* d_count = nullptr;
    int32_t* d_flags = nullptr;    // rdx_search_async(out_flags): the "incomplete" word of the packed partial, or NULL
    hipStream_t st = nullptr;
    unsigned long long seq = 0;
    bool exact_only = false, balance = false, ride = false, big_host_copy = false;
    int grid = 0, G = 0, nqt = 0, depth = 0;
    int64_t sample_rows = 0;
    double expected_per_query = 0.0;   // candidates per query a random corpus would have emitted (0: exact path)
    size_t b_s = 0, b_r = 0, b_c = 0;
    rdx_search_stats stats = {};   // rdx_search_async only: the statistics of the deferred search
};

// ------------------------------------------------------------------------------------------------
// the index: one corpus shard resident in one GPU's HBM
// ------------------------------------------------------------------------------------------------
struct rdx_index {
    int device = 0;
    int dim = 0, dim_pad = 0, ksteps = 0, scale_log2 = 0;
    int n_cu = 256;
    int64_t rows = 0, cap = 0;   // cap is a multiple of 256
    float* master = nullptr;     // [cap][dim] normalised fp32 rows (default), or NULL with option compact_master:
    uint16_t* raw16 = nullptr;   //   [cap][dim] raw bf16 rows as delivered ...
    double* den = nullptr;       //   [cap] ... and their divisors max(|x|, 1e-12); k_rows.hpp MasterView
    int compact = 0;             // option "compact_master" (settable while the index is empty): 4 instead of 6 B/element
    MasterView mv() const { return MasterView{master, raw16, den}; }
    _Float16* shadow = nullptr;  // [cap][dim_pad] fp16 scan copy in MFMA fragment order (rdx_common.hpp corpus_off)
    hipStream_t own_stream = nullptr;
    std::mutex mu;

    // options
    int force_exact = 0, force_fast = 0, profile = 0, sib_sync = 0, sib_lag = 6, retry = 1, xcd_balance = 1, fuse_epilogue = 1, force_bn = 0, wave_layout = 0;
    int half_boot = 1;       // option: 129..256 queries take their threshold sample as two 128-query tiles per sampled corpus tile
    int small_scan = 1;      // option: k_scan_small (split-K over all rows) as the main scan of small launches
    int split_boot = 1;      // option: k_boot (K loop split over the waves) for the threshold bootstrap of small launches
    int fuse_finish = 1;     // option: the end-of-search work runs in the last block of the search's last kernel (0: its own launch k_finish)
    int spec_tau = 1;        // option: speculative scan threshold (rank < k of the sample, verified by k_refine)
    int dense_sample = 0;    // searches left with a threshold sample twice as dense (set when a search emitted 3x a random corpus' candidates)
    int spread_boot = 1;     // option: the threshold sample is every div-th 32-row block instead of every div-th 256-row tile (B > 64)
    int spec_backoff = 0;    // searches left during which the provable threshold is used (set when a speculation failed)
    double xw[8] = {1, 1, 1, 1, 1, 1, 1, 1};   // relative speed of the XCDs as the last main scans showed it (sum 8)
    unsigned long long wg_times[1024] = {};    // start/end stamps of the last main scan's workgroups (host copy)
    int sample_div = 64;
    int64_t cand_cap = 0;   // 0 = automatic
    int64_t row_base = 0;   // added to every returned row id (global ids of a shard)
    int64_t* row_map = nullptr;   // [cap] local row -> returned row id (strictly increasing), or NULL = local + row_base

    // scratch (grow-only; never allocated inside a warmed-up search)
    DevBuf r_list, r_q, r_s, r_r, r_c;   // second-chance batch of overflowed queries
    DevBuf wgt;                          // [grid][2] workgroup time stamps of the main scan
    std::unordered_map<const void*, size_t> func_lds;   // dynamic-LDS limit already raised for a kernel ON THIS DEVICE
    DevBuf sib_scratch, staging, qraw, qhat, qshadow, tau, cntw, cand, setmax, exact_list, iota, dense, ctr, bad, o_score, o_row,
        o_count, mask, ids;
    // end-of-search mailbox in pinned host memory (k_finish writes it over PCIe; the host spins on its sequence number)
    Mailbox* mbox = nullptr;          // host address
    Mailbox* mbox_dev = nullptr;      // the same memory as the device sees it
    unsigned long long seq = 0;       // number of the last search enqueued on this index
    char* pin_out = nullptr;          // pinned staging for the small results of host callers (score | row | count)
    char* pin_out_dev = nullptr;
    size_t pin_out_bytes = 0;
    bool ctr_ready = false;           // the counter block was zeroed once; afterwards every k_finish re-zeroes it
    PendingSearch pending;            // rdx_search_async: the search whose host half is still to run
    hipEvent_t ev[8] = {};
    bool ev_ok = false;
    rdx_search_stats stats = {};

    float scale() const { return std::ldexp(1.0f, scale_log2); }
    float two_e() const { return 2.0f * (1.0e-3f + 2.5e-7f * (float)dim_pad); }   // see DESIGN.md "error bound"
};

// a `where` bitmap kept resident in HBM between searches (the reference's filters are a handful of fixed shapes)
struct rdx_mask {
    int device = 0;
    int64_t rows = 0;   // row count of the index when the mask was made: a mask never outlives a write to the index
    DevBuf words;
};

// a pinned, device-visible word the merge kernel publishes ((sequence << 1) | value) and the host waits on
struct rdx_signal {
    int device = 0;
    unsigned long long* host = nullptr;
    unsigned long long* dev = nullptr;
    unsigned long long seq = 0;   // number of the last merge that was given this signal
};

static int finish_pending(rdx_index* h, bool* redone);   // (search section)

static size_t shadow_bytes(const rdx_index* h, int64_t cap) { return (size_t)cap * h->dim_pad * 2; }

static int set_device(const rdx_index* h) {
    HIP_TRY(hipSetDevice(h->device));
    return RDX_OK;
}

static void free_master(float* m, uint16_t* r, double* d) {
    if (m) (void)hipFree(m);
    if (r) (void)hipFree(r);
    if (d) (void)hipFree(d);
}

// master storage for `cap` rows in the index's mode
static hipError_t alloc_master(const rdx_index* h, int64_t cap, float** m, uint16_t** r, double** d) {
    *m = nullptr;
    *r = nullptr;
    *d = nullptr;
    if (!h->compact) return hipMalloc((void**)m, (size_t)cap * h->dim * 4);
    hipError_t e = hipMalloc((void**)r, (size_t)cap * h->dim * 2);
    if (e == hipSuccess) e = hipMalloc((void**)d, (size_t)cap * 8);
    if (e != hipSuccess) {
        free_master(nullptr, *r, *d);
        *r = nullptr;
        *d = nullptr;
    }
    return e;
}

static int grow(rdx_index* h, int64_t need_rows) {
    if (need_rows <= h->cap) return RDX_OK;
    int64_t ncap = std::max<int64_t>(need_rows, h->cap + h->cap / 2);
    ncap = (ncap + 255) / 256 * 256;
    float* nm = nullptr;
    uint16_t* nr16 = nullptr;
    double* nd = nullptr;
    _Float16* ns = nullptr;
    hipError_t e = alloc_master(h, ncap, &nm, &nr16, &nd);
    if (e == hipSuccess) e = hipMalloc((void**)&ns, shadow_bytes(h, ncap));
    if (e != hipSuccess && ncap > (need_rows + 255) / 256 * 256) {   // retry without head-room
        free_master(nm, nr16, nd);
        ncap = (need_rows + 255) / 256 * 256;
        e = alloc_master(h, ncap, &nm, &nr16, &nd);
        if (e == hipSuccess) e = hipMalloc((void**)&ns, shadow_bytes(h, ncap));
    }
    if (e != hipSuccess) {
        free_master(nm, nr16, nd);
        return fail(RDX_ERR_NOMEM, std::string("growing index to ") + std::to_string(ncap) + " rows: " + hipGetErrorString(e));
    }
    hipStream_t st = h->own_stream;
    int64_t* nr = nullptr;
    if (h->row_map) {   // the id map grows with the rows: old entries kept, new ones start as local + row_base
        e = hipMalloc((void**)&nr, (size_t)ncap * 8);
        if (e != hipSuccess) {
            free_master(nm, nr16, nd);
            (void)hipFree(ns);
            return fail(RDX_ERR_NOMEM, std::string("growing the row id map: ") + hipGetErrorString(e));
        }
        if (h->rows > 0) HIP_TRY(hipMemcpyAsync(nr, h->row_map, (size_t)h->rows * 8, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_iota64, dim3((unsigned)((ncap - h->rows + 255) / 256)), dim3(256), 0, st, nr, h->rows, ncap - h->rows, h->row_base);
    }
    HIP_TRY(hipMemsetAsync(ns, 0, shadow_bytes(h, ncap), st));
    if (h->rows > 0) {
        if (h->compact) {
            HIP_TRY(hipMemcpyAsync(nr16, h->raw16, (size_t)h->rows * h->dim * 2, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(nd, h->den, (size_t)h->rows * 8, hipMemcpyDeviceToDevice, st));
        } else {
            HIP_TRY(hipMemcpyAsync(nm, h->master, (size_t)h->rows * h->dim * 4, hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(hipMemcpyAsync(ns, h->shadow, shadow_bytes(h, h->cap), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    free_master(h->master, h->raw16, h->den);
    if (h->shadow) (void)hipFree(h->shadow);
    if (h->row_map) (void)hipFree(h->row_map);
    h->master = nm;
    h->raw16 = nr16;
    h->den = nd;
    h->shadow = ns;
    h->row_map = nr;
    h->cap = ncap;
    return RDX_OK;
}

// ------------------------------------------------------------------------------------------------
// library / lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int rdx_version(void) { return RDX_ABI_VERSION; }
extern "C" const char* rdx_last_error(void) { return g_err.c_str(); }

extern "C" int rdx_device_count(int* n) {
    if (!n) return fail(RDX_ERR_INVALID, "rdx_device_count: null pointer");
    HIP_TRY(hipGetDeviceCount(n));
    return RDX_OK;
}

static int check_dim(int dim) {
    if (dim <= 0 || dim % 4 != 0 || dim > MAX_DIM)
        return fail(RDX_ERR_INVALID, "dim must be a positive multiple of 4, at most " + std::to_string(MAX_DIM) + " (got " +
                                         std::to_string(dim) + ")");
    return RDX_OK;
}

extern "C" int rdx_index_create(int device, int dim, rdx_index** out) {
    if (!out) return fail(RDX_ERR_INVALID, "rdx_index_create: null out pointer");
    RDX_TRY(check_dim(dim));
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(RDX_ERR_INVALID, "device " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RDX_ERR_STATE, std::string("librdx is built for gfx950 (MI355X) only; device reports ") + prop.gcnArchName);
    rdx_index* h = new rdx_index();
    h->device = device;
    h->dim = dim;
    h->dim_pad = (dim + 63) / 64 * 64;
    h->ksteps = h->dim_pad / 64;
    h->scale_log2 = (int)std::lround(std::log2(std::sqrt((double)dim)));
    h->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(RDX_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = h;
    return RDX_OK;
}

extern "C" int rdx_index_destroy(rdx_index* h) {
    if (!h) return RDX_OK;
    (void)hipSetDevice(h->device);
    if (h->pending.active) (void)hipStreamSynchronize(h->pending.st);   // an abandoned asynchronous search: let its kernels finish
    (void)hipStreamSynchronize(h->own_stream);
    free_master(h->master, h->raw16, h->den);
    if (h->shadow) (void)hipFree(h->shadow);
    if (h->row_map) (void)hipFree(h->row_map);
    for (DevBuf* b : {&h->staging, &h->qraw, &h->qhat, &h->qshadow, &h->tau, &h->cntw, &h->cand, &h->setmax, &h->exact_list,
                      &h->iota, &h->dense, &h->ctr, &h->bad, &h->o_score, &h->o_row, &h->o_count, &h->mask, &h->ids,
                      &h->sib_scratch, &h->r_list, &h->r_q, &h->r_s, &h->r_r, &h->r_c})
        b->release();
    if (h->mbox) (void)hipHostFree(h->mbox);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->ev_ok)
        for (auto& e : h->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(h->own_stream);
    delete h;
    return RDX_OK;
}

extern "C" int rdx_index_dim(const rdx_index* h, int* dim) {
    if (!h || !dim) return fail(RDX_ERR_INVALID, "rdx_index_dim: null pointer");
    *dim = h->dim;
    return RDX_OK;
}

extern "C" int rdx_index_count(const rdx_index* h, int64_t* rows) {
    if (!h || !rows) return fail(RDX_ERR_INVALID, "rdx_index_count: null pointer");
    *rows = h->rows;
    return RDX_OK;
}

extern "C" int rdx_index_reserve(rdx_index* h, int64_t rows) {
    if (!h || rows < 0) return fail(RDX_ERR_INVALID, "rdx_index_reserve: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(set_device(h));
    return grow(h, rows);
}

extern "C" int rdx_index_set_option(rdx_index* h, const char* name, int64_t value) {
    if (!h || !name) return fail(RDX_ERR_INVALID, "rdx_index_set_option: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));   // the host half of an asynchronous search reads the options it was enqueued under
    const std::string n(name);
    if (n == "force_exact") h->force_exact = value != 0;
    else if (n == "force_fast") h->force_fast = value != 0;
    else if (n == "sib_sync") h->sib_sync = value != 0;
    else if (n == "retry") h->retry = value != 0;
    else if (n == "fuse_epilogue") h->fuse_epilogue = value != 0;
    else if (n == "wave_layout") h->wave_layout = value == 1 ? 1 : 0;
    else if (n == "fuse_finish") h->fuse_finish = value != 0;
    else if (n == "split_boot") h->split_boot = value != 0;
    else if (n == "small_scan") h->small_scan = value != 0;
    else if (n == "half_boot") h->half_boot = value != 0;
    else if (n == "spread_boot") h->spread_boot = value != 0;
    else if (n == "spec_tau") {
        h->spec_tau = value != 0;
        h->spec_backoff = 0;
    }
    else if (n == "force_bn") {
        if (value != 0 && value != 64 && value != 128 && value != 256) return fail(RDX_ERR_INVALID, "force_bn must be 0 (automatic), 64, 128 or 256");
        h->force_bn = (int)value;
    }
    else if (n == "compact_master") {
        if (h->rows > 0 || h->cap > 0) return fail(RDX_ERR_STATE, "compact_master can only be chosen while the index is empty");
        h->compact = value != 0;
    }
    else if (n == "xcd_balance") {
        h->xcd_balance = value != 0;
        for (double& w : h->xw) w = 1.0;
    }
    else if (n == "sib_lag") h->sib_lag = (int)std::min<int64_t>(std::max<int64_t>(value, 3), 100);
    else if (n == "profile") h->profile = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 3);
    else if (n == "sample_div") {
        if (value < 1) return fail(RDX_ERR_INVALID, "sample_div must be >= 1");
        h->sample_div = (int)std::min<int64_t>(value, 1 << 20);
    } else if (n == "row_base") {
        if (value < 0) return fail(RDX_ERR_INVALID, "row_base must be >= 0");
        h->row_base = value;
    } else if (n == "cand_cap") {
        if (value < 0) return fail(RDX_ERR_INVALID, "cand_cap must be 0 (auto) or a positive slot count per (query, stream) segment");
        h->cand_cap = value;
    } else
        return fail(RDX_ERR_INVALID, "unknown option '" + n + "'");
    return RDX_OK;
}

extern "C" int rdx_index_xcd_shares(rdx_index* h, double* out8, const double* in8) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_index_xcd_shares: null index");
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    if (in8) {
        double w[8], sum = 0;
        for (int x = 0; x < 8; ++x) {
            if (!(in8[x] > 0.0) || !(in8[x] < 100.0)) return fail(RDX_ERR_INVALID, "rdx_index_xcd_shares: shares must be positive finite numbers");
            sum += (w[x] = std::min(1.5, std::max(0.6, in8[x])));
        }
        for (int x = 0; x < 8; ++x) h->xw[x] = w[x] * 8.0 / sum;
    }
    if (out8)
        for (int x = 0; x < 8; ++x) out8[x] = h->xw[x];
    return RDX_OK;
}

// ------------------------------------------------------------------------------------------------
// ingest
// ------------------------------------------------------------------------------------------------
static const int64_t STAGE_ROWS = 32768;

// normalise n rows (host or device, fp32 or bf16) into master/shadow at dst rows (row0.. or dst_ids)
static int ingest(rdx_index* h, const void* rows, bool is_bf16, int64_t n, int space, int64_t row0, const int64_t* d_dst_ids,
                  bool verbatim = false) {
    hipStream_t st = h->own_stream;
    if (space == RDX_DEVICE) HIP_TRY(hipDeviceSynchronize());   // the caller's producers of `rows` (any stream) are done
    const size_t esz = is_bf16 ? 2 : 4;
    RDX_TRY(h->bad.ensure(sizeof(int)));
    HIP_TRY(hipMemsetAsync(h->bad.p, 0, sizeof(int), st));
    for (int64_t off = 0; off < n; off += STAGE_ROWS) {
        const int64_t m = std::min(STAGE_ROWS, n - off);
        const char* src = reinterpret_cast<const char*>(rows) + (size_t)off * h->dim * esz;
        if (space == RDX_HOST) {
            RDX_TRY(h->staging.ensure((size_t)STAGE_ROWS * h->dim * 4));
            HIP_TRY(hipMemcpyAsync(h->staging.p, src, (size_t)m * h->dim * esz, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));   // pageable source: keep the copy ordered with the caller's buffer
            src = h->staging.as<char>();
        }
        const int grid = (int)((m + 3) / 4);
        hipLaunchKernelGGL(k_normalize<false>, dim3(grid), dim3(256), 0, st, is_bf16 ? nullptr : (const float*)src,
                           is_bf16 ? (const uint16_t*)src : nullptr, m, h->dim, d_dst_ids ? d_dst_ids + off : nullptr,
                           row0 + off, h->mv(), h->shadow, h->ksteps, h->scale(), h->bad.as<int>(), (int64_t)0, verbatim ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (space == RDX_HOST) HIP_TRY(hipStreamSynchronize(st));   // staging is reused by the next chunk
    }
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad) return fail(RDX_ERR_INVALID, "embeddings contain NaN or Inf");
    return RDX_OK;
}

static int add_impl(rdx_index* h, const void* rows, bool is_bf16, int64_t n, int space, bool verbatim = false) {
    if (!h || (n > 0 && !rows) || n < 0) return fail(RDX_ERR_INVALID, "rdx_index_add: bad argument");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (n == 0) return RDX_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    if (h->compact && (!is_bf16 || verbatim))
        return fail(RDX_ERR_STATE, "this index keeps a compact master (raw bf16 rows + divisors): rows must arrive through rdx_index_add_bf16");
    RDX_TRY(set_device(h));
    RDX_TRY(grow(h, h->rows + n));
    RDX_TRY(ingest(h, rows, is_bf16, n, space, h->rows, nullptr, verbatim));
    h->rows += n;
    return RDX_OK;
}

extern "C" int rdx_index_add(rdx_index* h, const float* rows, int64_t n, int space) { return add_impl(h, rows, false, n, space); }
extern "C" int rdx_index_add_bf16(rdx_index* h, const uint16_t* rows, int64_t n, int space) {
    return add_impl(h, rows, true, n, space);
}
extern "C" int rdx_index_add_stored(rdx_index* h, const float* rows, int64_t n, int space) {
    return add_impl(h, rows, false, n, space, true);
}

// copy a host or device int64 id list to the device scratch `ids`, validating on the host when possible
static int stage_ids(rdx_index* h, const int64_t* ids, int64_t n, int space, const int64_t** d_ids) {
    hipStream_t st = h->own_stream;
    std::vector<int64_t> tmp;
    const int64_t* host_ids = ids;
    if (space == RDX_DEVICE) {
        HIP_TRY(hipDeviceSynchronize());   // the caller's producers of `ids` (any stream) are done
        tmp.resize((size_t)n);
        HIP_TRY(hipMemcpy(tmp.data(), ids, (size_t)n * 8, hipMemcpyDeviceToHost));
        host_ids = tmp.data();
    }
    for (int64_t i = 0; i < n; ++i)
        if (host_ids[i] < 0 || host_ids[i] >= h->rows)
            return fail(RDX_ERR_INVALID, "row id " + std::to_string(host_ids[i]) + " out of range [0, " + std::to_string(h->rows) + ")");
    if (space == RDX_DEVICE) {
        *d_ids = ids;
        return RDX_OK;
    }
    RDX_TRY(h->ids.ensure((size_t)n * 8));
    HIP_TRY(hipMemcpyAsync(h->ids.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    *d_ids = h->ids.as<int64_t>();
    return RDX_OK;
}

extern "C" int rdx_index_update(rdx_index* h, const int64_t* row_ids, const float* rows, int64_t n, int space) {
    if (!h || n < 0 || (n > 0 && (!row_ids || !rows))) return fail(RDX_ERR_INVALID, "rdx_index_update: bad argument");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (n == 0) return RDX_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    if (h->compact) return fail(RDX_ERR_STATE, "rdx_index_update takes fp32 rows: not available on an index with a compact (bf16) master");
    RDX_TRY(set_device(h));
    const int64_t* d_ids = nullptr;
    RDX_TRY(stage_ids(h, row_ids, n, space, &d_ids));
    return ingest(h, rows, false, n, space, 0, d_ids);
}

extern "C" int rdx_index_get(rdx_index* h, const int64_t* row_ids, int64_t n, float* out, int space) {
    if (!h || n < 0 || (n > 0 && (!row_ids || !out))) return fail(RDX_ERR_INVALID, "rdx_index_get: bad argument");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (n == 0) return RDX_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    RDX_TRY(set_device(h));
    hipStream_t st = h->own_stream;
    const int64_t* d_ids = nullptr;
    RDX_TRY(stage_ids(h, row_ids, n, space, &d_ids));
    for (int64_t off = 0; off < n; off += STAGE_ROWS) {
        const int64_t m = std::min(STAGE_ROWS, n - off);
        float* dst = out + (size_t)off * h->dim;
        if (space == RDX_HOST) {
            RDX_TRY(h->staging.ensure((size_t)STAGE_ROWS * h->dim * 4));
            dst = h->staging.as<float>();
        }
        hipLaunchKernelGGL(k_gather_rows, dim3((int)((m + 3) / 4)), dim3(256), 0, st, h->mv(), d_ids + off, m, h->dim, dst);
        HIP_TRY(hipGetLastError());
        if (space == RDX_HOST) {
            HIP_TRY(hipMemcpyAsync(out + (size_t)off * h->dim, dst, (size_t)m * h->dim * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    return RDX_OK;
}

extern "C" int rdx_index_compact(rdx_index* h, const int64_t* keep, int64_t n_keep) {
    if (!h || n_keep < 0 || (n_keep > 0 && !keep)) return fail(RDX_ERR_INVALID, "rdx_index_compact: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    RDX_TRY(set_device(h));
    for (int64_t i = 0; i < n_keep; ++i) {
        if (keep[i] < 0 || keep[i] >= h->rows) return fail(RDX_ERR_INVALID, "compact: row id out of range");
        if (i > 0 && keep[i] <= keep[i - 1]) return fail(RDX_ERR_INVALID, "compact: keep list must be strictly ascending");
    }
    hipStream_t st = h->own_stream;
    const int64_t ncap = std::max<int64_t>(256, (n_keep + 255) / 256 * 256);
    float* nm = nullptr;
    uint16_t* nr16 = nullptr;
    double* nd = nullptr;
    _Float16* ns = nullptr;
    hipError_t e = alloc_master(h, ncap, &nm, &nr16, &nd);
    if (e == hipSuccess) e = hipMalloc((void**)&ns, shadow_bytes(h, ncap));
    if (e != hipSuccess) {
        free_master(nm, nr16, nd);
        return fail(RDX_ERR_NOMEM, std::string("compact: ") + hipGetErrorString(e));
    }
    HIP_TRY(hipMemsetAsync(ns, 0, shadow_bytes(h, ncap), st));
    if (n_keep > 0) {
        RDX_TRY(h->ids.ensure((size_t)n_keep * 8));
        HIP_TRY(hipMemcpyAsync(h->ids.p, keep, (size_t)n_keep * 8, hipMemcpyHostToDevice, st));
        const MasterView nv{nm, nr16, nd};
        if (h->compact)
            hipLaunchKernelGGL(k_gather_raw, dim3((int)((n_keep + 3) / 4)), dim3(256), 0, st, h->mv(), h->ids.as<int64_t>(), n_keep, h->dim, nv);
        else
            hipLaunchKernelGGL(k_gather_rows, dim3((int)((n_keep + 3) / 4)), dim3(256), 0, st, h->mv(), h->ids.as<int64_t>(), n_keep, h->dim, nm);
        hipLaunchKernelGGL(k_reshadow, dim3((int)((n_keep + 3) / 4)), dim3(256), 0, st, nv, (int64_t)0, n_keep, h->dim, ns,
                           h->ksteps, h->scale());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    free_master(h->master, h->raw16, h->den);
    if (h->shadow) (void)hipFree(h->shadow);
    if (h->row_map) (void)hipFree(h->row_map);   // rows were renumbered: the caller sets a new id map (or none)
    h->row_map = nullptr;
    h->master = nm;
    h->raw16 = nr16;
    h->den = nd;
    h->shadow = ns;
    h->cap = ncap;
    h->rows = n_keep;
    return RDX_OK;
}

extern "C" int rdx_index_set_row_ids(rdx_index* h, int64_t first_row, const int64_t* ids, int64_t n, int space) {
    if (!h || first_row < 0 || n < 0 || (n > 0 && !ids)) return fail(RDX_ERR_INVALID, "rdx_index_set_row_ids: bad argument");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    std::lock_guard<std::mutex> lk(h->mu);
    if (first_row + n > h->rows) return fail(RDX_ERR_INVALID, "rdx_index_set_row_ids: rows [first_row, first_row + n) must exist");
    if (n == 0) return RDX_OK;
    RDX_TRY(finish_pending(h, nullptr));   // k_refine / k_select_dense of an asynchronous search may still be reading the map
    RDX_TRY(set_device(h));
    hipStream_t st = h->own_stream;
    std::vector<int64_t> tmp;
    const int64_t* host_ids = ids;
    if (space == RDX_DEVICE) {
        HIP_TRY(hipDeviceSynchronize());   // the caller's producer of `ids` (any stream) is done
        tmp.resize((size_t)n);             // the merge's tie order rests on the map being increasing: checked for device ids too
        HIP_TRY(hipMemcpy(tmp.data(), ids, (size_t)n * 8, hipMemcpyDeviceToHost));
        host_ids = tmp.data();
    }
    for (int64_t i = 0; i < n; ++i)
        if (host_ids[i] < 0 || (i > 0 && host_ids[i] <= host_ids[i - 1]))
            return fail(RDX_ERR_INVALID, "rdx_index_set_row_ids: ids must be non-negative and strictly increasing");
    if (!h->row_map) {
        HIP_TRY(hipMalloc((void**)&h->row_map, (size_t)std::max<int64_t>(h->cap, 256) * 8));
        hipLaunchKernelGGL(k_iota64, dim3((unsigned)((h->cap + 255) / 256)), dim3(256), 0, st, h->row_map, (int64_t)0, h->cap, h->row_base);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h->row_map + first_row, ids, (size_t)n * 8, space == RDX_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RDX_OK;
}

// scratch of rdx_l2_normalize, kept per device (the call sits on the embed() path of every query: no hipMalloc/hipFree per call)
struct NormScratch {
    std::mutex mu;
    DevBuf in, out, bad;
};
// (never destroyed: a static destructor would call hipFree at process exit, possibly after the HIP runtime is gone)
static NormScratch* const g_norm = new NormScratch[64];

extern "C" int rdx_enc_attention_f16(int device, const void* qkv, const int32_t* tok_first, const int32_t* tok_len, int64_t n_tokens,
                                     int heads, int head_dim, float scale, int max_text_tokens, void* ctx, void* stream) {
    if (n_tokens < 0 || heads < 1 || heads > 65535) return fail(RDX_ERR_INVALID, "rdx_enc_attention_f16: bad shape");
    if (head_dim != ENC_HEAD_DIM) return fail(RDX_ERR_INVALID, "rdx_enc_attention_f16: head_dim must be 64");
    if (n_tokens == 0) return RDX_OK;
    if (!qkv || !tok_first || !tok_len || !ctx) return fail(RDX_ERR_INVALID, "rdx_enc_attention_f16: null pointer");
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_attention_f16: qkv and ctx must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_attention_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    // keys a workgroup stages in LDS: its 64 tokens' texts span at most 64 + 2 (L - 1) tokens when no text is longer than L
    int window = ENC_KEY_WINDOW;
    if (max_text_tokens > 0) window = std::min<int64_t>(ENC_KEY_WINDOW, (64 + 2 * ((int64_t)max_text_tokens - 1) + 7) / 8 * 8);
    hipLaunchKernelGGL(k_enc_attention, dim3((unsigned)((n_tokens + 63) / 64), (unsigned)heads), dim3(256), (size_t)window * 256, (hipStream_t)stream,
                       (const _Float16*)qkv, tok_first, tok_len, n_tokens, heads, scale, window, (_Float16*)ctx);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

extern "C" int rdx_enc_attention_mfma_f16(int device, const void* qkv, const int32_t* query_blocks, int n_blocks, int heads, int head_dim,
                                          float scale, void* ctx, void* stream) {
    if (n_blocks < 0 || heads < 1 || heads > 65535) return fail(RDX_ERR_INVALID, "rdx_enc_attention_mfma_f16: bad shape");
    if (head_dim != ENC_HEAD_DIM) return fail(RDX_ERR_INVALID, "rdx_enc_attention_mfma_f16: head_dim must be 64");
    if (n_blocks == 0) return RDX_OK;
    if (!qkv || !query_blocks || !ctx) return fail(RDX_ERR_INVALID, "rdx_enc_attention_mfma_f16: null pointer");
    if (((uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)query_blocks) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_attention_mfma_f16: pointers must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_attention_mfma_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_enc_attention_mfma, dim3((unsigned)n_blocks, (unsigned)heads), dim3(256), 0, (hipStream_t)stream, (const _Float16*)qkv, query_blocks, heads,
                       scale * 1.4426950408889634f, (_Float16*)ctx);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

template <bool GELU>
static void launch_linear_small(int ntb, dim3 grid, hipStream_t st, const _Float16* x, const _Float16* w, const _Float16* b, int T, int N, int K,
                                _Float16* out) {
    if (ntb <= 2 && K % 1024 == 0 && K <= 4096) {   // one question: 16 waves split K, a wave's whole slice in flight at once
        const size_t lds = (size_t)16 * ntb * 1024;
        if (ntb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<1, GELU, 16>), grid, dim3(1024), lds, st, x, w, b, T, N, K, out);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<2, GELU, 16>), grid, dim3(1024), lds, st, x, w, b, T, N, K, out);
        return;
    }
    const size_t lds = (size_t)ntb * 4096;   // 4 waves x ntb * 4 registers x 64 lanes x 4 B
    switch (ntb) {
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<1, GELU, 4>), grid, dim3(256), lds, st, x, w, b, T, N, K, out); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<2, GELU, 4>), grid, dim3(256), lds, st, x, w, b, T, N, K, out); break;
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<4, GELU, 4>), grid, dim3(256), lds, st, x, w, b, T, N, K, out); break;
        case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<8, GELU, 4>), grid, dim3(256), lds, st, x, w, b, T, N, K, out); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_enc_linear_small<16, GELU, 4>), grid, dim3(256), lds, st, x, w, b, T, N, K, out); break;
    }
}

extern "C" int rdx_enc_linear_small_f16(int device, const void* x, const void* w, const void* bias, int n_tokens, int n_out, int n_in,
                                        int act, void* out, void* stream) {
    if (n_tokens < 0 || n_tokens > 256) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: at most 256 tokens (use the BLAS library beyond)");
    if (n_out < 16 || n_out % 16 || n_in < 512 || n_in % 512) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: n_out must be a multiple of 16, n_in of 512");
    if (act != 0 && act != 1) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: act is 0 (none) or 1 (erf GELU)");
    if (n_tokens == 0) return RDX_OK;
    if (!x || !w || !bias || !out) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: null pointer");
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: x, w and out must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_linear_small_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    int ntb = 1;
    while (ntb * 16 < n_tokens) ntb *= 2;
    const dim3 grid((unsigned)(n_out / 16));
    if (act) launch_linear_small<true>(ntb, grid, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, n_tokens, n_out, n_in, (_Float16*)out);
    else launch_linear_small<false>(ntb, grid, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, n_tokens, n_out, n_in, (_Float16*)out);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

extern "C" int rdx_enc_add_layernorm_f16(int device, const void* a, const void* b, const void* gamma, const void* beta, float eps,
                                         int64_t rows, int hidden, void* out, void* stream) {
    if (rows < 0 || hidden < 512 || hidden > 2048 || hidden % 512) return fail(RDX_ERR_INVALID, "rdx_enc_add_layernorm_f16: hidden must be 512, 1024, 1536 or 2048");
    if (rows == 0) return RDX_OK;
    if (!a || !b || !gamma || !beta || !out) return fail(RDX_ERR_INVALID, "rdx_enc_add_layernorm_f16: null pointer");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15)
        return fail(RDX_ERR_INVALID, "rdx_enc_add_layernorm_f16: pointers must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_add_layernorm_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const _Float16 *pa = (const _Float16*)a, *pb = (const _Float16*)b, *pg = (const _Float16*)gamma, *pbt = (const _Float16*)beta;
    switch (hidden / 512) {
        case 1: hipLaunchKernelGGL(k_enc_add_ln<1>, grid, block, 0, st, pa, pb, pg, pbt, eps, rows, (_Float16*)out); break;
        case 2: hipLaunchKernelGGL(k_enc_add_ln<2>, grid, block, 0, st, pa, pb, pg, pbt, eps, rows, (_Float16*)out); break;
        case 3: hipLaunchKernelGGL(k_enc_add_ln<3>, grid, block, 0, st, pa, pb, pg, pbt, eps, rows, (_Float16*)out); break;
        default: hipLaunchKernelGGL(k_enc_add_ln<4>, grid, block, 0, st, pa, pb, pg, pbt, eps, rows, (_Float16*)out); break;
    }
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

// ---- the single-question forward: E4..E7 (enc_small.hpp) ----------------------------------------------------------------------
// kernels with more than 64 KiB of dynamic LDS need the limit raised once per kernel and device
static std::mutex g_enc_attr_mu;
static std::unordered_map<const void*, size_t>* const g_enc_attr = new std::unordered_map<const void*, size_t>[64];
static int enc_dynamic_lds(int device, const void* func, size_t bytes) {
    if (bytes <= 65536) return RDX_OK;
    std::lock_guard<std::mutex> lk(g_enc_attr_mu);
    size_t& have = g_enc_attr[device][func];
    if (have < bytes) {
        HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return RDX_OK;
}

template <int NTB, int KCS, int NPH, int FPB, bool LNPRO, int EPI>
static int launch_enc_stage(int device, const EncStage& a, hipStream_t st) {
    const size_t lds = (size_t)NTB * 16384 + (size_t)NTB * 16 * KCS * 512 * 2;
    auto* fn = k_enc_stage<NTB, KCS, NPH, FPB, LNPRO, EPI>;
    RDX_TRY(enc_dynamic_lds(device, (const void*)fn, lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)(a.N / FPB)), dim3(1024), lds, st, a);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

template <int NTB, int KCS, int NPH>
static int dispatch_enc_stage(int device, const EncStage& a, bool lnpro, int epi, int fpb, hipStream_t st) {
    if (lnpro) {
        if constexpr (NPH == 1) {
            if (epi == ENC_EPI_BIAS) return launch_enc_stage<NTB, KCS, 1, 16, true, ENC_EPI_BIAS>(device, a, st);
            if (epi == ENC_EPI_GELU) return launch_enc_stage<NTB, KCS, 1, 16, true, ENC_EPI_GELU>(device, a, st);
        }
        return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: the LayerNorm prologue takes n_in 512 or 1024 and epilogue 0 or 1");
    }
#define RDX_ENC_PLAIN(F)                                                                                                   \
    if (fpb == F) {                                                                                                        \
        if (epi == ENC_EPI_BIAS) return launch_enc_stage<NTB, KCS, NPH, F, false, ENC_EPI_BIAS>(device, a, st);            \
        if (epi == ENC_EPI_GELU) return launch_enc_stage<NTB, KCS, NPH, F, false, ENC_EPI_GELU>(device, a, st);            \
        return launch_enc_stage<NTB, KCS, NPH, F, false, ENC_EPI_RESIDUAL>(device, a, st);                                  \
    }
    RDX_ENC_PLAIN(16)
    RDX_ENC_PLAIN(8)
    RDX_ENC_PLAIN(4)
#undef RDX_ENC_PLAIN
    return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: features_per_workgroup is 16, 8 or 4");
}

extern "C" int rdx_enc_stage_f16(int device, const void* x, const int64_t* x_rows, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                 void* y_out, const void* w, const void* bias, const void* res, int n_tokens, int n_out, int n_in,
                                 int epilogue, int features_per_workgroup, void* out, void* stream) {
    if (n_tokens < 0 || n_tokens > 32) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: at most 32 tokens");
    if (n_in != 512 && n_in != 1024 && n_in != 2048 && n_in != 4096) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: n_in must be 512, 1024, 2048 or 4096");
    if (epilogue < 0 || epilogue > 2) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: epilogue is 0 (bias), 1 (bias + erf GELU) or 2 (bias + residual)");
    int fpb = features_per_workgroup ? features_per_workgroup : 16;
    if (n_out < fpb || n_out % fpb) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: n_out must be a multiple of features_per_workgroup");
    const bool lnpro = ln_gamma != nullptr;
    if (lnpro && (!ln_beta || x_rows)) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: the LayerNorm prologue needs gamma and beta and takes no row list");
    if (lnpro && fpb != 16) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: the LayerNorm prologue runs with 16 features per workgroup");
    if (epilogue == 2 && !res) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: epilogue 2 needs the residual");
    if (n_tokens == 0) return RDX_OK;
    if (!x || !w || !bias || !out) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: null pointer");
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)y_out | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15)
        return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: pointers must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_stage_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    EncStage a;
    a.x = (const _Float16*)x;
    a.x_rows = x_rows;
    a.gamma = (const _Float16*)ln_gamma;
    a.beta = (const _Float16*)ln_beta;
    a.eps = ln_eps;
    a.y_out = (_Float16*)y_out;
    a.w = (const _Float16*)w;
    a.bias = (const _Float16*)bias;
    a.res = (const _Float16*)res;
    a.out = (_Float16*)out;
    a.T = n_tokens;
    a.N = n_out;
    hipStream_t st = (hipStream_t)stream;
#define RDX_ENC_K(NTB)                                                                         \
    switch (n_in) {                                                                            \
        case 512: return dispatch_enc_stage<NTB, 1, 1>(device, a, lnpro, epilogue, fpb, st);   \
        case 1024: return dispatch_enc_stage<NTB, 2, 1>(device, a, lnpro, epilogue, fpb, st);  \
        case 2048: return dispatch_enc_stage<NTB, 2, 2>(device, a, lnpro, epilogue, fpb, st);  \
        default: return dispatch_enc_stage<NTB, 2, 4>(device, a, lnpro, epilogue, fpb, st);    \
    }
    if (n_tokens <= 16) { RDX_ENC_K(1) }
    RDX_ENC_K(2)
#undef RDX_ENC_K
}

extern "C" int rdx_enc_attention_small_f16(int device, const void* qkv, const int32_t* tok_first, int n_tokens, int heads, int head_dim,
                                           float scale, void* ctx, void* stream) {
    if (n_tokens < 0 || n_tokens > 32) return fail(RDX_ERR_INVALID, "rdx_enc_attention_small_f16: at most 32 tokens");
    if (heads < 1 || heads > 65535 || head_dim != ENC_HEAD_DIM) return fail(RDX_ERR_INVALID, "rdx_enc_attention_small_f16: head_dim must be 64");
    if (n_tokens == 0) return RDX_OK;
    if (!qkv || !tok_first || !ctx) return fail(RDX_ERR_INVALID, "rdx_enc_attention_small_f16: null pointer");
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_attention_small_f16: qkv and ctx must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_attention_small_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    const float sl2 = scale * 1.4426950408889634f;
    if (n_tokens <= 16)
        hipLaunchKernelGGL(k_enc_attn_small<1>, dim3((unsigned)heads), dim3(64), 0, (hipStream_t)stream, (const _Float16*)qkv, tok_first, n_tokens, heads, sl2, (_Float16*)ctx);
    else
        hipLaunchKernelGGL(k_enc_attn_small<2>, dim3((unsigned)heads), dim3(128), 0, (hipStream_t)stream, (const _Float16*)qkv, tok_first, n_tokens, heads, sl2, (_Float16*)ctx);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

extern "C" int rdx_enc_embed_f16(int device, const int64_t* tok, const int64_t* pos_id, const void* word, const void* pos, const void* type0,
                                 int n_tokens, int hidden, void* out, void* stream) {
    if (n_tokens < 0 || hidden < 512 || hidden > 2048 || hidden % 512) return fail(RDX_ERR_INVALID, "rdx_enc_embed_f16: hidden must be 512, 1024, 1536 or 2048");
    if (n_tokens == 0) return RDX_OK;
    if (!tok || !pos_id || !word || !pos || !type0 || !out) return fail(RDX_ERR_INVALID, "rdx_enc_embed_f16: null pointer");
    if (((uintptr_t)word | (uintptr_t)pos | (uintptr_t)type0 | (uintptr_t)out) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_embed_f16: pointers must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_embed_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    const dim3 grid((unsigned)((n_tokens + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const _Float16 *pw = (const _Float16*)word, *pp = (const _Float16*)pos, *pt = (const _Float16*)type0;
    switch (hidden / 512) {
        case 1: hipLaunchKernelGGL(k_enc_embed<1>, grid, block, 0, st, tok, pos_id, pw, pp, pt, n_tokens, (_Float16*)out); break;
        case 2: hipLaunchKernelGGL(k_enc_embed<2>, grid, block, 0, st, tok, pos_id, pw, pp, pt, n_tokens, (_Float16*)out); break;
        case 3: hipLaunchKernelGGL(k_enc_embed<3>, grid, block, 0, st, tok, pos_id, pw, pp, pt, n_tokens, (_Float16*)out); break;
        default: hipLaunchKernelGGL(k_enc_embed<4>, grid, block, 0, st, tok, pos_id, pw, pp, pt, n_tokens, (_Float16*)out); break;
    }
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

extern "C" int rdx_enc_layernorm_rows_f16(int device, const void* s, const void* gamma, const void* beta, float eps, int rows, int hidden,
                                          float* out, void* stream) {
    if (rows < 0 || hidden < 512 || hidden > 2048 || hidden % 512) return fail(RDX_ERR_INVALID, "rdx_enc_layernorm_rows_f16: hidden must be 512, 1024, 1536 or 2048");
    if (rows == 0) return RDX_OK;
    if (!s || !gamma || !beta || !out) return fail(RDX_ERR_INVALID, "rdx_enc_layernorm_rows_f16: null pointer");
    if (((uintptr_t)s | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) return fail(RDX_ERR_INVALID, "rdx_enc_layernorm_rows_f16: pointers must be 16-byte aligned");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_layernorm_rows_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const _Float16 *ps = (const _Float16*)s, *pg = (const _Float16*)gamma, *pb = (const _Float16*)beta;
    switch (hidden / 512) {
        case 1: hipLaunchKernelGGL(k_enc_ln_rows<1>, grid, block, 0, st, ps, pg, pb, eps, rows, out); break;
        case 2: hipLaunchKernelGGL(k_enc_ln_rows<2>, grid, block, 0, st, ps, pg, pb, eps, rows, out); break;
        case 3: hipLaunchKernelGGL(k_enc_ln_rows<3>, grid, block, 0, st, ps, pg, pb, eps, rows, out); break;
        default: hipLaunchKernelGGL(k_enc_ln_rows<4>, grid, block, 0, st, ps, pg, pb, eps, rows, out); break;
    }
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}


extern "C" int rdx_enc_gelu_f16(int device, void* x, int64_t n, void* stream) {
    if (n < 0 || n % 8) return fail(RDX_ERR_INVALID, "rdx_enc_gelu_f16: n must be a non-negative multiple of 8");
    if (n == 0) return RDX_OK;
    if (!x || ((uintptr_t)x & 15)) return fail(RDX_ERR_INVALID, "rdx_enc_gelu_f16: x must be a 16-byte aligned device pointer");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_enc_gelu_f16: device out of range");
    HIP_TRY(hipSetDevice(device));
    const int64_t n8 = n / 8;
    const unsigned grid = (unsigned)std::min<int64_t>((n8 + 255) / 256, 2048);
    hipLaunchKernelGGL(k_enc_gelu, dim3(grid), dim3(256), 0, (hipStream_t)stream, (_Float16*)x, n8);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}


extern "C" int rdx_l2_normalize(int device, const float* in, int64_t n, int dim, float* out, int space, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(RDX_ERR_INVALID, "rdx_l2_normalize: bad argument");
    RDX_TRY(check_dim(dim));
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (device < 0 || device >= 64) return fail(RDX_ERR_INVALID, "rdx_l2_normalize: device out of range");
    if (n == 0) return RDX_OK;
    HIP_TRY(hipSetDevice(device));
    NormScratch& sc = g_norm[device];
    std::lock_guard<std::mutex> lk(sc.mu);
    hipStream_t st = (hipStream_t)stream;
    const float* d_in = in;
    float* d_out = out;
    RDX_TRY(sc.bad.ensure(sizeof(int)));
    if (space == RDX_HOST) {
        RDX_TRY(sc.in.ensure((size_t)n * dim * 4));
        RDX_TRY(sc.out.ensure((size_t)n * dim * 4));
        HIP_TRY(hipMemcpyAsync(sc.in.p, in, (size_t)n * dim * 4, hipMemcpyHostToDevice, st));
        d_in = sc.in.as<float>();
        d_out = sc.out.as<float>();
    }
    HIP_TRY(hipMemsetAsync(sc.bad.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_normalize<false>, dim3((int)((n + 3) / 4)), dim3(256), 0, st, d_in, (const uint16_t*)nullptr, n, dim,
                       (const int64_t*)nullptr, (int64_t)0, MasterView{d_out, nullptr, nullptr}, (_Float16*)nullptr, 0, 1.0f, sc.bad.as<int>());
    HIP_TRY(hipGetLastError());
    int b = 0;
    if (space == RDX_HOST) HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * dim * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&b, sc.bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));   // the NaN/Inf verdict is part of the return value
    if (b) return fail(RDX_ERR_INVALID, "embeddings contain NaN or Inf");
    return RDX_OK;
}

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
// kernels using more than 64 KiB of dynamic LDS need the limit raised once per kernel and device (the attribute is
// per device: the cache lives in the index, which is bound to one)
static int ensure_dynamic_lds(rdx_index* h, const void* func, size_t bytes) {
    size_t& have = h->func_lds[func];
    if (have < bytes) {
        HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return RDX_OK;
}

template <int BN, int EPI, bool RES, bool SIBT = false, bool NTT = false, bool FUSED = false>
static int launch_scan(rdx_index* h, const ScanParams& p, int grid, hipStream_t st) {
    // LDS: query-image ring (or the whole resident query tile) + BN hit counters + BN thresholds
    const size_t lds = (size_t)(RES ? p.ksteps : RING_SLOTS) * BN * BK * 2 + BN * 8;
    void (*kern)(const ScanParams) = p.allow ? k_scan<BN, EPI, true, RES, SIBT, NTT, FUSED> : k_scan<BN, EPI, false, RES, SIBT, NTT, FUSED>;
    RDX_TRY(ensure_dynamic_lds(h, (const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

// developer experiment (option "wave_layout" = 1): the B > 128 fused main scan with one wave per SIMD, csrc/scan_w4.hpp
static int launch_scan_w4(rdx_index* h, const ScanParams& p, int grid, hipStream_t st) {
    const size_t lds = (size_t)RING_SLOTS * 256 * BK * 2 + 256 * 8 + 256 * 64;   // ring + counters + thresholds + the emit path's staging rows
    void (*kern)(const ScanParams) = p.allow ? k_scan_w4<true> : k_scan_w4<false>;
    RDX_TRY(ensure_dynamic_lds(h, (const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}

template <int EPI>
static int launch_scan_bn(rdx_index* h, int bn, bool res, const ScanParams& p, int grid, hipStream_t st) {
    // NTT (5th template argument): one query tile -> every corpus byte is read by exactly one workgroup -> non-temporal loads
    if (bn == 64) {
        if (p.nqt == 1) return res ? launch_scan<64, EPI, true, false, true>(h, p, grid, st) : launch_scan<64, EPI, false, false, true>(h, p, grid, st);
        return res ? launch_scan<64, EPI, true>(h, p, grid, st) : launch_scan<64, EPI, false>(h, p, grid, st);
    }
    if (bn == 128) return p.nqt == 1 ? launch_scan<128, EPI, false, false, true>(h, p, grid, st) : launch_scan<128, EPI, false>(h, p, grid, st);
    if constexpr (EPI == EPI_EMIT) {
#ifdef RDX_CHECK_BOUNDS
        constexpr bool HAVE_FUSED = false;   // the address-checking test build carries one more live value: no fused variants
#else
        constexpr bool HAVE_FUSED = true;
#endif
        // option fuse_epilogue (default on: +1 % at B = 1024 since the static wave priority went in): with an even
        // number of k-steps per tile the emit check of a tile rides with the first k-step of the next one
        if constexpr (HAVE_FUSED) {
            if ((p.ksteps & 1) == 0 && h->fuse_epilogue) {
                if (p.sib) return launch_scan<256, EPI, false, true, false, true>(h, p, grid, st);
                if (p.nqt == 1) return launch_scan<256, EPI, false, false, true, true>(h, p, grid, st);
                if (h->wave_layout == 1 && p.tile_stride == 1) return launch_scan_w4(h, p, grid, st);
                return launch_scan<256, EPI, false, false, false, true>(h, p, grid, st);
            }
        }
        if (p.sib) return launch_scan<256, EPI, false, true>(h, p, grid, st);
        if (p.nqt == 1) return launch_scan<256, EPI, false, false, true>(h, p, grid, st);   // one query tile: corpus read once -> nt loads
        return launch_scan<256, EPI, false>(h, p, grid, st);
    } else {
        if (p.nqt == 1) return launch_scan<256, EPI, false, false, true>(h, p, grid, st);
        return launch_scan<256, EPI, false>(h, p, grid, st);
    }
}

static const int K_FAST_MAX = 256;   // larger k goes through the exact full scan

// exact full scan for the queries listed in d_list[0..n_list)
static int run_exact(rdx_index* h, const int32_t* d_list, int n_list, int k, const uint32_t* d_allow, float* d_score,
                     int64_t* d_row, int32_t* d_count, hipStream_t st, bool stamps = false, const FinishArgs* fin = nullptr) {
    const FinishArgs no_fin = {};   // (ctr == NULL: the launch does not end a search)
    // profile = 3: the scoring kernel's first block and the select kernel's last block leave their times in the counter block
    unsigned long long* t_first = stamps ? reinterpret_cast<unsigned long long*>(h->ctr.as<char>() + offsetof(RefineCounters, t_first_inv)) : nullptr;
    unsigned long long* t_last = stamps ? reinterpret_cast<unsigned long long*>(h->ctr.as<char>() + offsetof(RefineCounters, t_last)) : nullptr;
    RDX_TRY(h->dense.ensure((size_t)QX * std::max<int64_t>(h->rows, 1) * 4));
    const int grid_rows = (int)std::min<int64_t>((h->rows + 3) / 4, (int64_t)h->n_cu * 16);   // one row per wave up to 16 Ki rows
    for (int j0 = 0; j0 < n_list; j0 += QX) {
        const int nq = std::min(QX, n_list - j0);
        if (h->rows > 0 && h->dim <= 1024) {
            // queries in registers, two rows in flight per wave, 2 blocks per CU (all resident at once)
            const int u = (h->dim / 4 + 63) / 64;
            const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((h->rows + 3) / 4, (int64_t)h->n_cu * 2))), b(256);
#define RDX_K5A(U) hipLaunchKernelGGL(k_exact_scores_reg<U>, g, b, 0, st, h->mv(), h->rows, h->dim, h->qhat.as<float>(), d_list + j0, nq, d_allow, h->dense.as<float>(), t_first)
            if (u == 1) RDX_K5A(1);
            else if (u == 2) RDX_K5A(2);
            else if (u == 3) RDX_K5A(3);
            else RDX_K5A(4);
#undef RDX_K5A
            HIP_TRY(hipGetLastError());
        } else if (h->rows > 0) {
            hipLaunchKernelGGL(k_exact_scores, dim3(std::max(grid_rows, 1)), dim3(256), (size_t)nq * h->dim * 4, st, h->mv(),
                               h->rows, h->dim, h->qhat.as<float>(), d_list + j0, nq, d_allow, h->dense.as<float>(), t_first);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_select_dense, dim3(nq), dim3(1024), 0, st, h->dense.as<float>(), h->rows, d_list + j0, k, h->row_base,
                           h->row_map, d_score, d_row, d_count, t_last, (fin && j0 + QX >= n_list) ? *fin : no_fin);
        HIP_TRY(hipGetLastError());
    }
    return RDX_OK;
}

// Host callers (HostOut): small results travel with the end-of-search kernel into pinned staging and are copied to the
// caller's buffers by the CPU once the mailbox says the search is complete; large ones use D2H copies. When a fallback pass
// had to rewrite some results afterwards they are copied again (HostOut::stale).
static const size_t PIN_MAX = 256 * 1024;   // results up to this size ride with k_finish (one block writing over PCIe)

static int ensure_mailbox(rdx_index* h) {
    if (h->mbox) return RDX_OK;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(p, 0, sizeof(Mailbox));
    void* d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, p, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(p);
        return fail(RDX_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    h->mbox = reinterpret_cast<Mailbox*>(p);
    h->mbox_dev = reinterpret_cast<Mailbox*>(d);
    return RDX_OK;
}

static int ensure_pin_out(rdx_index* h, size_t bytes) {
    if (bytes <= h->pin_out_bytes) return RDX_OK;
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    h->pin_out = nullptr;
    h->pin_out_bytes = 0;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, PIN_MAX, hipHostMallocMapped | hipHostMallocCoherent));
    void* d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, p, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(p);
        return fail(RDX_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    h->pin_out = reinterpret_cast<char*>(p);
    h->pin_out_dev = reinterpret_cast<char*>(d);
    h->pin_out_bytes = PIN_MAX;
    return RDX_OK;
}

// The search numbered `seq` has completed: its k_finish published the mailbox. Spin on the pinned word for a while
// (short searches: the store arrives a couple of us after the kernel, no interrupt, no D2H copy), then fall back to
// hipStreamSynchronize (long searches; it also surfaces a faulted kernel).
// Waits until ready() says a word in pinned host memory has arrived — never for the stream: an asynchronous caller may have enqueued
// other work behind the search or merge that publishes the word (BASELINE config 5: the next batch's query encode, 15 ms of kernels),
// and a stream synchronise would wait for that too. Hot spin for 0.4 ms (a small search ends inside it), then poll with a yield
// between looks (a 0.6 - 25 ms search is noticed within a microsecond; measured with 20 us sleeps instead: +30 us on a 0.56 ms search,
// +170 us on a 2.2 ms one), after 200 ms with 50 us sleeps. The stream is only QUERIED, every 50 ms, to turn a failed or vanished
// launch into an error instead of an endless wait. 0 = arrived, 1 = the stream ran dry without the word, < 0 = error code.
// Wait policy (rdx_set_wait_policy; environment RDX_WAIT_SPIN_US / RDX_WAIT_SLEEP_US at load): the default burns a host core for the
// length of a search — right for a benchmark or one rank per GPU, wrong for a server whose sessions share the cores (the reference
// serves concurrent Streamlit sessions from one process, app.py:42-43). sleep_us > 0: after the hot spin the waiter SLEEPS that long
// between looks (a 15 ms scan then costs the core ~1 % instead of 100 %; the result is noticed up to sleep_us later).
static std::atomic<int> g_wait_spin_us{[] {
    const char* e = std::getenv("RDX_WAIT_SPIN_US");
    return e ? std::atoi(e) : 400;
}()};
static std::atomic<int> g_wait_sleep_us{[] {
    const char* e = std::getenv("RDX_WAIT_SLEEP_US");
    return e ? std::max(0, std::atoi(e)) : 0;
}()};
extern "C" int rdx_set_wait_policy(int spin_us, int sleep_us) {
    if (spin_us < 0 || sleep_us < 0 || sleep_us > 1000000) return fail(RDX_ERR_INVALID, "rdx_set_wait_policy: spin_us >= 0, 0 <= sleep_us <= 1000000");
    g_wait_spin_us.store(spin_us);
    g_wait_sleep_us.store(sleep_us);
    return RDX_OK;
}
template <class F>
static int wait_word(F ready, hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    const int spin_us = g_wait_spin_us.load(std::memory_order_relaxed), sleep_us = g_wait_sleep_us.load(std::memory_order_relaxed);
    for (unsigned spins = 1;; ++spins) {
        if (ready()) return 0;
        if (spin_us == 0) break;
        _mm_pause();
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    auto next_query = t0 + std::chrono::milliseconds(50);
    const auto t_sleep = t0 + std::chrono::milliseconds(200);
    for (unsigned n = 1;; ++n) {
        if (ready()) return 0;
        if (sleep_us > 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
            if (ready()) return 0;
        } else if ((n & 15u) != 0) {
            sched_yield();
            continue;
        }
        const auto now = std::chrono::steady_clock::now();
        if (sleep_us == 0 && now >= t_sleep) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (now >= next_query) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) return ready() ? 0 : 1;   // everything enqueued has run: the word must be there
            if (e != hipErrorNotReady) return fail(RDX_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
            next_query = now + std::chrono::milliseconds(50);
        }
    }
}

static int wait_search(rdx_index* h, hipStream_t st, unsigned long long seq) {
    const int rc = wait_word([&] { return __atomic_load_n(&h->mbox->seq, __ATOMIC_ACQUIRE) == seq; }, st);
    if (rc == 1) return fail(RDX_ERR_HIP, "internal: the search completed without publishing its mailbox");
    return rc;
}

// depth 0 = the caller's batch; depth 1 = the second-chance batch of queries whose candidate segments overflowed
static int complete_chunk_impl(rdx_index* h, const PendingSearch& ps, rdx_search_stats* acc_stats, HostOut* ho, bool* redone);
// The counter block is zeroed once and afterwards by the k_finish of every search. A search that leaves early (allocation
// failure, launch error, an internal check) may have skipped its k_finish: the next search zeroes the block itself again.
static int complete_chunk(rdx_index* h, const PendingSearch& ps, rdx_search_stats* acc_stats, HostOut* ho, bool* redone) {
    const int rc = complete_chunk_impl(h, ps, acc_stats, ho, redone);
    if (rc != RDX_OK) h->ctr_ready = false;
    return rc;
}

// `defer`: return as soon as everything up to k_finish is enqueued; the host half is left in h->pending (rdx_search_async)
static int search_chunk_impl(rdx_index* h, const float* d_queries, int64_t nq, int k, const uint32_t* d_allow, float* d_score,
                             int64_t* d_row, int32_t* d_count, hipStream_t st, rdx_search_stats* acc_stats, int depth,
                             HostOut* ho, bool defer, int32_t* d_flags);
static int search_chunk(rdx_index* h, const float* d_queries, int64_t nq, int k, const uint32_t* d_allow, float* d_score,
                        int64_t* d_row, int32_t* d_count, hipStream_t st, rdx_search_stats* acc_stats, int depth = 0,
                        HostOut* ho = nullptr, bool defer = false, int32_t* d_flags = nullptr) {
    const int rc = search_chunk_impl(h, d_queries, nq, k, d_allow, d_score, d_row, d_count, st, acc_stats, depth, ho, defer, d_flags);
    if (rc != RDX_OK) h->ctr_ready = false;
    return rc;
}

static int search_chunk_impl(rdx_index* h, const float* d_queries, int64_t nq, int k, const uint32_t* d_allow, float* d_score,
                             int64_t* d_row, int32_t* d_count, hipStream_t st, rdx_search_stats* acc_stats, int depth,
                             HostOut* ho, bool defer, int32_t* d_flags) {
    const int nq_pad = (int)((nq + 255) / 256 * 256);
    const bool prof_all = h->profile == 1 && depth == 0, prof_main = (h->profile == 1 || h->profile == 2) && depth == 0;
    const bool prof_stamps = h->profile == 3 && depth == 0;   // the kernels stamp their own times: nothing extra on the stream
    auto mark = [&](int i) {
        if (prof_all || (prof_main && (i == 3 || i == 4))) (void)hipEventRecord(h->ev[i], st);
    };
    double ps_expected_per_query = 0.0;   // candidates per query a random corpus would emit with this search's sample (MFMA path)
    // K1 on the queries: qhat (fp32, exact re-score) + tiled fp16 copy (scan)
    RDX_TRY(h->qhat.ensure((size_t)nq_pad * h->dim * 4));
    RDX_TRY(h->qshadow.ensure((size_t)nq_pad * h->dim_pad * 2));
    constexpr size_t SIB_OFF = 64, SIB_BYTES = (size_t)REFINE_STREAMS * 16;   // counters | sibling progress bytes
    static_assert(sizeof(RefineCounters) <= SIB_OFF, "counter block layout");
    RDX_TRY(h->ctr.ensure(SIB_OFF + SIB_BYTES));
    RDX_TRY(h->exact_list.ensure((size_t)nq_pad * 4));
    RDX_TRY(ensure_mailbox(h));
    if (!h->ctr_ready) {   // zeroed once; afterwards the k_finish of every search leaves it zeroed for the next one
        HIP_TRY(hipMemsetAsync(h->ctr.p, 0, SIB_OFF + SIB_BYTES, st));
        h->ctr_ready = true;
    }
    int* d_bad = reinterpret_cast<int*>(h->ctr.as<char>() + offsetof(RefineCounters, bad));
    const unsigned long long seq = ++h->seq;
    mark(0);
    hipLaunchKernelGGL(k_normalize<true>, dim3((int)((nq_pad + 3) / 4)), dim3(256), 0, st, d_queries, (const uint16_t*)nullptr, nq, h->dim,
                       (const int64_t*)nullptr, (int64_t)0, MasterView{h->qhat.as<float>(), nullptr, nullptr}, h->qshadow.as<_Float16>(), h->ksteps, h->scale(),
                       d_bad, (int64_t)nq_pad, depth > 0 ? 1 : 0);   // depth 1: the rows ARE normalised queries (gathered from qhat): kept bit for bit
    HIP_TRY(hipGetLastError());
    mark(1);

    // small problems and huge k are served by the exact full scan alone (one fp32 read of the corpus)
    // Round 2 re-measured the crossover (B = 4: exact path 0.060 / 0.099 / 0.129 ms at 24 k / 40 k / 60 k rows, MFMA path 0.081 / 0.085 /
    // 0.087): the exact path costs ceil(nq / 4) passes of (1.28 us per 1000 rows + 10 us) on top of what both paths share, the
    // MFMA path ~60 us more than that share whatever the size — and beyond 32 Ki rows the select no longer holds a score row in
    // registers. (The rule it replaces, nq * rows <= 4 M below 64 Ki rows, sent 64 queries x 60 k rows through 16 exact passes.)
    const int64_t exact_passes = (nq + 3) / 4;
    const bool small = h->rows <= 32768 && (double)exact_passes * ((double)h->rows * 1.28e-3 + 10.0) <= 60.0;   // (any size: 600 queries on 1000 rows are 150 passes)
    const bool exact_only = h->force_exact || k > K_FAST_MAX || k == 0 || h->rows < 1 || (small && !h->force_fast);
    int64_t sample_rows = 0;
    int grid = 0, G = 0, nqt = 0;
    bool balance = false;
    // K6 (end of search): results of small host calls -> pinned staging, counters (+ workgroup stamps) -> mailbox, counter block
    // re-zeroed, sequence number published. Runs in the last block of the search's last kernel (option fuse_finish, default) or
    // as its own launch behind it.
    const size_t b_s = (size_t)nq * k * 4, b_r = (size_t)nq * k * 8, b_c = (size_t)nq * 4;
    const bool ride = ho && b_s + b_r + b_c <= PIN_MAX;
    if (ride) RDX_TRY(ensure_pin_out(h, b_s + b_r + b_c));
    auto finish_args = [&](bool stamps) {
        FinishArgs f = {};
        f.ctr = h->ctr.as<RefineCounters>();
        f.mb = h->mbox_dev;
        f.seq = seq;
        f.wgt = stamps ? h->wgt.as<unsigned long long>() : nullptr;
        f.n_wgt = stamps ? 2 * grid : 0;
        f.s0 = reinterpret_cast<const uint32_t*>(d_row);
        f.d0 = reinterpret_cast<uint32_t*>(h->pin_out_dev);
        f.w0 = (int64_t)(ride ? b_r / 4 : 0);
        f.s1 = reinterpret_cast<const uint32_t*>(d_score);
        f.d1 = reinterpret_cast<uint32_t*>(h->pin_out_dev + b_r);
        f.w1 = (int64_t)(ride ? b_s / 4 : 0);
        f.s2 = reinterpret_cast<const uint32_t*>(d_count);
        f.d2 = reinterpret_cast<uint32_t*>(h->pin_out_dev + b_r + b_s);
        f.w2 = (int64_t)(ride ? b_c / 4 : 0);
        f.out_flags = depth == 0 ? d_flags : nullptr;
        f.may_redo = (!exact_only && depth == 0) ? 1 : 0;
        return f;
    };
    const FinishArgs no_fin = {};
    FinishArgs fin_later = {};   // fuse_finish = 0: what the stand-alone k_finish gets
    if (exact_only) {
        for (int i = 2; i <= 3; ++i) mark(i);   // events 3..4 bracket the dominant kernels of this path too (K5a + K5b)
        if ((size_t)nq_pad * 4 > h->iota.bytes) {   // identity query list, uploaded once (grow-only), not per search
            RDX_TRY(h->iota.ensure((size_t)nq_pad * 4));
            const size_t cnt = h->iota.bytes / 4;
            std::vector<int32_t> io(cnt);
            for (size_t i = 0; i < cnt; ++i) io[i] = (int32_t)i;
            HIP_TRY(hipMemcpyAsync(h->iota.p, io.data(), cnt * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        fin_later = finish_args(false);
        RDX_TRY(run_exact(h, h->iota.as<int32_t>(), (int)nq, k, d_allow, d_score, d_row, d_count, st, prof_stamps,
                          h->fuse_finish ? &fin_later : nullptr));
        for (int i = 4; i <= 5; ++i) mark(i);
    } else {
        // queries per workgroup: 64 (tile resident in LDS), 128, 256. 257..384 queries run as three 128-query tiles rather than
        // one full and one half-empty 256-query tile (measured at 1M x 1024, B = 384: 0.78 vs 0.84 ms; tools/bn_sweep.py)
        int bn = nq <= 64 ? 64 : (nq <= 128 ? 128 : ((nq > 256 && nq <= 384) ? 128 : 256));
        if (h->force_bn && (nq + h->force_bn - 1) / h->force_bn <= 32) bn = h->force_bn;   // developer option: queries per workgroup
        nqt = (int)((nq + bn - 1) / bn);
        grid = std::max(8, h->n_cu / 8 * 8);
        const int wpx = grid / 8;
        if (nqt > wpx) return fail(RDX_ERR_STATE, "internal: query chunk larger than one scan launch");
        G = wpx / nqt;
        const int n_streams = 8 * G;
        if (n_streams > REFINE_STREAMS) return fail(RDX_ERR_STATE, "internal: more streams than the refine kernel gathers");
        const int n_sets = n_streams * SETS_PER_STREAM;
        const int64_t n_tiles = (h->rows + 255) / 256;
        if (n_tiles * h->ksteps >= ((int64_t)1 << 31)) return fail(RDX_ERR_STATE, "shard too large for one scan launch");
        // the 64-query tile stays resident in LDS when all its k-step images fit (no DMA, no barrier in the main loop)
        const bool res = bn == 64 && (size_t)h->ksteps * 8192 + 512 <= 160 * 1024 - 1024;
        // bootstrap sample: every div-th tile. More rows sampled = tighter tau = fewer hits; keep the expected hits
        // per query (~1.3 k rows/sample_rows) around 4000/... of the refine list and the sample >= max(64k, 8192) rows
        // Bootstrap geometry. 129..256 queries run their main scan as ONE 256-query tile per workgroup, but their bootstrap samples
        // ~130 tiles: as one tile per workgroup that is half the CUs working through 16 dependent k-steps of 64 KB each (36 us at
        // c3). As TWO 128-query tiles per sampled tile every CU works, a k-step moves 48 KB and takes 1.4 instead of 2.25 us
        // (DESIGN.md §10's table): option "half_boot" (default 1).
        int bn_b = bn, nqt_b = nqt, ns_b = n_streams;
        if (h->half_boot && bn == 256 && nqt == 1) {
            bn_b = 128;
            nqt_b = 2;
            ns_b = 8 * (wpx / 2);
        }
        const int n_sets_b = ns_b * SETS_PER_STREAM;
        const int64_t want_rows = std::max<int64_t>(64 * (int64_t)k, 8192);
        int div = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(h->sample_div, h->rows / want_rows), 3000 / std::max(k, 1)));
        if (depth > 0) div = std::max(1, div / 8);   // second chance: 8x denser sample -> a threshold that sees the cluster
        // A corpus whose last searches emitted far more candidates than a random corpus would (clustered rows: a query's neighbours are
        // one document's chunks, and a thin sample holds too few of them to place the threshold among them) gets twice the sample for a
        // while: +0.25 ms of bootstrap on a 10 M-row scan, against thousands of surplus candidates per query to gather and re-score
        // (measured, embedding-like corpus at c4: 19.6 -> 16.2 ms per batch; N(0,1) corpus: +1 %, which is why it is not the default).
        else if (h->dense_sample > 0 && div > 1) div = std::max(1, div / 2);
        int64_t n_sched = (n_tiles + div - 1) / div;
        // whole rounds only: the bootstrap takes as long as its busiest stream, so 77 tiles on 64 streams cost two tiles' time for
        // 1.2 tiles' worth of threshold (a 1.25 M-row shard at B = 1024: 53 -> 27 us of a 2.26 ms search); thin the sample to the
        // last full round instead, as long as it keeps the rows asked for above
        if (n_sched > ns_b && n_sched % ns_b != 0) {
            const int64_t full = n_sched / ns_b * ns_b;
            const int div2 = (int)((n_tiles + full - 1) / full);
            if ((n_tiles + div2 - 1) / div2 * 256 >= want_rows) {
                div = div2;
                n_sched = (n_tiles + div - 1) / div;
            }
        }
        // The sample as every div-th 32-ROW BLOCK (option "spread_boot", default 1) instead of every div-th 256-row tile: the same number
        // of rows, eight times finer. Wave w of virtual tile j takes block (8 j + w) * div; the last virtual tile ends inside the corpus.
        const int64_t n_blocks32 = (h->rows + 31) / 32;
        const int64_t n_virtual = ((n_blocks32 - 1) / div + 1) / 8;
        const bool spread = h->spread_boot && n_virtual >= 1;
        if (spread) n_sched = n_virtual;
        sample_rows = n_sched * 256;
        int n_sets_used = (int)std::min<int64_t>(ns_b, n_sched) * SETS_PER_STREAM;
        // Small launches (<= 64 queries and a sample of at most four 32-row blocks per CU): the split-K bootstrap k_boot — one
        // 32-row block per workgroup, the k-steps dealt to the waves — instead of a few whole tiles of 16 dependent k-steps on
        // a few CUs (scan_kernel.hpp K2b). Whole rounds of the CUs when more than one.
        int64_t boot_units = std::min<int64_t>(n_blocks32, n_sched * 8);
        if (boot_units > h->n_cu) boot_units = boot_units / h->n_cu * h->n_cu;
        const bool use_boot = h->split_boot && bn == BOOT_BN && nqt == 1 && boot_units <= 4 * (int64_t)h->n_cu;
        // ... and the split-K main scan k_scan_small when the whole corpus is at most 32 such blocks per CU (scan_kernel.hpp K2c)
        const bool use_small = h->small_scan && bn == BOOT_BN && nqt == 1 && h->ksteps <= 16 && n_streams == grid &&
                               n_blocks32 <= 32 * (int64_t)h->n_cu && n_blocks32 >= grid;
        int boot_sets = 0;
        if (use_boot) {
            sample_rows = boot_units * 32;
            boot_sets = (int)boot_units * 4;
            n_sets_used = boot_sets;
        }
        // slots per (query, stream) segment: 8x the expected hits, power of two, [32, 4096]
        const double exp_hits = (1.5 * k * (double)h->rows / (double)std::max<int64_t>(sample_rows, 1) + k) / n_streams;
        ps_expected_per_query = exp_hits * n_streams;
        // (slots cost address space, not bandwidth: only occupied slots are ever touched)
        // (nq_pad * n_streams is 65,536 whatever the batch: 1024 slots = 512 MiB, 4096 = 2 GiB of the 288)
        uint32_t capw = depth > 0 ? 4096 : 1024;
        while (capw < 4096 && capw < 8.0 * exp_hits) capw *= 2;
        if (h->cand_cap && depth == 0) capw = (uint32_t)std::min<int64_t>(h->cand_cap, 8191);
        // the scan addresses candidate slots with 32-bit indices (scan_kernel.hpp emit_block)
        if ((uint64_t)nq_pad * (uint64_t)n_streams * capw >= (1ull << 29)) return fail(RDX_ERR_STATE, "internal: candidate segments exceed the 32-bit slot index");
        RDX_TRY(h->tau.ensure((size_t)nq_pad * 4));
        RDX_TRY(h->cntw.ensure((size_t)nq_pad * n_streams * 4));
        RDX_TRY(h->cand.ensure((size_t)nq_pad * n_streams * capw * 8));
        RDX_TRY(h->setmax.ensure((size_t)nq_pad * std::max(std::max(n_sets, n_sets_b), boot_sets) * 4));

        ScanParams p = {};
        p.shadow = h->shadow;
        p.qshadow = h->qshadow.as<_Float16>();
        p.ksteps = h->ksteps;
        p.rows = h->rows;
        p.n_tiles = n_tiles;
        p.nqt = nqt;
        p.nq_pad = nq_pad;
        p.allow = d_allow;
        p.setmax = h->setmax.as<float>();
        p.n_sets = n_sets;
        p.tau = h->tau.as<float>();
        p.cntw = h->cntw.as<uint32_t>();
        p.cand = h->cand.as<uint2>();
        p.capw = capw;
        p.inv_scale2 = std::ldexp(1.0f, -2 * h->scale_log2);
        p.sib_lag = h->sib_lag;
        p.shadow_bytes = (int64_t)shadow_bytes(h, h->cap);
        p.oob = reinterpret_cast<int*>(h->ctr.as<char>() + offsetof(RefineCounters, oob));
        RDX_TRY(h->sib_scratch.ensure((size_t)REFINE_STREAMS * 16 * 8));
        p.sib_scratch = h->sib_scratch.as<uint8_t>();
        p.sib = (nqt > 1 && nqt <= 16 && h->ksteps >= 4 && h->sib_sync) ? reinterpret_cast<uint32_t*>(h->ctr.as<char>() + SIB_OFF) : nullptr;
        if (p.sib) HIP_TRY(hipMemsetAsync(h->ctr.as<char>() + SIB_OFF, 0, SIB_BYTES, st));   // sibling progress bytes (option sib_sync only)

        if (use_boot) {
            BootParams bp = {};
            bp.shadow = h->shadow;
            bp.qshadow = h->qshadow.as<_Float16>();
            bp.ksteps = h->ksteps;
            bp.rows = h->rows;
            bp.n_blocks32 = n_blocks32;
            bp.units = (int)boot_units;
            bp.allow = d_allow;
            bp.setmax = h->setmax.as<float>();
            bp.n_sets = boot_sets;
            hipLaunchKernelGGL(k_boot, dim3((unsigned)boot_units), dim3(512), 0, st, bp);
            HIP_TRY(hipGetLastError());
        } else {
            ScanParams pb = p;
            pb.tile_stride = div;
            pb.span = TILE_ROWS;
            if (spread) {   // wave w of sampled entry j: block (8 j + w) * div (scan_kernel.hpp ScanParams::wave_off)
                pb.wave_off = (int64_t)(div - 1) * h->ksteps * 4096;
                pb.row_off = (div - 1) * 32;
                pb.span = (7 * div + 1) * 32;
                pb.n_tiles = (n_virtual - 1) * (int64_t)div + 1;   // ceil(n_tiles / div) = n_virtual entries: the last block lies inside the corpus
            }
            pb.nqt = nqt_b;
            pb.n_sets = n_sets_b;
            pb.sib = nullptr;
            RDX_TRY(launch_scan_bn<EPI_SETMAX>(h, bn_b, bn_b == bn ? res : false, pb, grid, st));
        }
        mark(2);
        // Speculative threshold. The provable threshold is the k-th largest sampled score: k/S of the sample's
        // quantile scale where the corpus' k-th score sits at k/N — with a 1.6 % sample and k = 10 that is 60x the hits one
        // needs. The corpus' k-th score is ESTIMATED by the sample's j-th largest with j ~ k*S/N; taking the smallest j for
        // which fewer than k rows of the corpus lie above it (with a factor 2 for the 2E band the verification needs) with
        // probability <= 1e-7 per query (the count above the sample's j-th largest is N/S * Gamma(j)) cuts the hits 2-6x
        // (c4: 890 -> ~430 per query, c3: 4500 -> ~700). k_refine verifies every query (c_k - 2E >= T); a failed one takes the
        // fallback passes, which use rank k, and switches speculation off for the next searches (structured corpora, where
        // "every div-th tile" is not a random sample).
        int k_sel = k;
        if (h->spec_tau && depth == 0 && h->spec_backoff == 0 && k > 1) {
            const double lam = 2.0 * (double)k * (double)sample_rows / (double)std::max<int64_t>(h->rows, 1);
            double term = std::exp(-lam), cdf = term;   // P(Poisson(lam) <= j - 1)
            int j = 1;
            while (1.0 - cdf > 1e-7 && j < k) {
                term *= lam / j;
                cdf += term;
                ++j;
            }
            k_sel = std::min(k, j);
        }
        if (depth == 0 && h->spec_backoff > 0) --h->spec_backoff;
        acc_stats->tau_rank = (float)k_sel;
        // proven threshold: 2E below the k-th sampled score — plus, when the sample was summed in another order than the main scan
        // sums (k_boot), twice the fp32 accumulation bound, so that the verification (c_k - 2E >= T, with c_k from the main
        // scan's sums) cannot fail on a rounding difference between the two orders
        const float slack = h->two_e() + ((use_boot != use_small) ? 2.0f * (float)h->dim_pad * 1.1920929e-7f : 0.0f);
        hipLaunchKernelGGL(k_tau, dim3(nq_pad), dim3(256), 0, st, h->setmax.as<float>(), use_boot ? boot_sets : n_sets_b, n_sets_used, k_sel,
                           k_sel == k ? slack * std::ldexp(1.0f, 2 * h->scale_log2) : 0.0f, (int)nq, h->tau.as<float>());
        HIP_TRY(hipGetLastError());
        mark(3);
        p.tile_stride = 1;
        // The eight XCDs do not finish equal shares at the same time (measured on c4: the last XCD 1.1-1.7 ms after the
        // first of 16.5, always the same ones). Each XCD therefore gets a contiguous range of the tile schedule sized by
        // its speed in the previous main scans (from the workgroups' own time stamps, damped) — no coordination
        // inside the kernel, just a different static split. Large launches only.
        balance = h->xcd_balance && depth == 0 && n_tiles >= 1024 && grid <= 512;
        if (balance) {
            RDX_TRY(h->wgt.ensure((size_t)grid * 16));
            p.use_xlo = 1;
            p.wgt = h->wgt.as<unsigned long long>();
            // bulk: what the slowest XCD should get, dealt interleaved to everybody (whole iterations of all streams);
            // tail: the rest, one contiguous range per XCD holding what that XCD should get beyond the bulk
            const double wmin = *std::min_element(h->xw, h->xw + 8);
            p.bulk_it = (int)std::max<int64_t>(0, (int64_t)std::floor((double)n_tiles * wmin / 8.0 / G) - 1);
            const int64_t t0 = (int64_t)p.bulk_it * n_streams, tail = n_tiles - t0;
            double want[8], sum = 0;
            for (int x = 0; x < 8; ++x) sum += (want[x] = std::max(0.0, (double)n_tiles * h->xw[x] / 8.0 - (double)p.bulk_it * G));
            double acc = 0;
            for (int x = 0; x <= 8; ++x) {
                p.xlo[x] = (int)(t0 + std::llround((double)tail * (sum > 0 ? acc / sum : x / 8.0)));
                if (x < 8) acc += want[x];
            }
            p.xlo[8] = (int)n_tiles;
        }
        if (prof_stamps && !balance && grid <= 512) {   // (Mailbox::wg_times holds 1024 stamps)
            RDX_TRY(h->wgt.ensure((size_t)grid * 16));
            p.wgt = h->wgt.as<unsigned long long>();
        }
        if (use_small) {
            SmallScanParams sp = {};
            sp.shadow = h->shadow;
            sp.qshadow = h->qshadow.as<_Float16>();
            sp.ksteps = h->ksteps;
            sp.rows = h->rows;
            sp.n_blocks32 = n_blocks32;
            sp.allow = d_allow;
            sp.tau = h->tau.as<float>();
            sp.cntw = h->cntw.as<uint32_t>();
            sp.cand = h->cand.as<uint2>();
            sp.capw = capw;
            sp.inv_scale2 = p.inv_scale2;
            sp.wgt = p.wgt;
            hipLaunchKernelGGL(k_scan_small, dim3((unsigned)grid), dim3(512), 0, st, sp);
            HIP_TRY(hipGetLastError());
        } else {
            RDX_TRY(launch_scan_bn<EPI_EMIT>(h, bn, res, p, grid, st));
        }
        p.use_xlo = 0;
        p.wgt = nullptr;
        mark(4);
        {
            // LDS list of the gathered hits: 16x the expected count (heavy-tailed score distributions of structured corpora; a list overflow costs a second pass), at most REFINE_LIST
            uint32_t list_cap = 1024;
            while (list_cap < (uint32_t)REFINE_LIST && list_cap < 16.0 * exp_hits * n_streams) list_cap *= 2;
            list_cap = std::min<uint32_t>(list_cap, REFINE_LIST);
            const size_t lds = (size_t)list_cap * 8;
            fin_later = finish_args((balance || prof_stamps) && grid <= 512);
            RDX_TRY(ensure_dynamic_lds(h, (const void*)k_refine, lds));
            hipLaunchKernelGGL(k_refine, dim3((int)nq), dim3(1024), lds, st, h->cand.as<uint2>(), h->cntw.as<uint32_t>(), n_streams, capw,
                               list_cap, k, h->two_e(), h->qhat.as<float>(), h->mv(), h->dim, h->row_base, h->row_map, d_score, d_row, d_count,
                               h->exact_list.as<int32_t>(), h->ctr.as<RefineCounters>(), h->tau.as<float>(), p.inv_scale2,
                               h->fuse_finish ? fin_later : no_fin);
            HIP_TRY(hipGetLastError());
        }
        mark(5);
    }
    if (!h->fuse_finish) {
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, st, fin_later);
        HIP_TRY(hipGetLastError());
    }
    if (ho && !ride) {   // large host results: plain copies behind the last kernel (complete_chunk synchronises the stream for them)
        if (k > 0) {
            HIP_TRY(hipMemcpyAsync(ho->score, d_score, b_s, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(ho->row, d_row, b_r, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipMemcpyAsync(ho->count, d_count, b_c, hipMemcpyDeviceToHost, st));
    }
    PendingSearch ps;
    ps.active = true;
    ps.d_queries = d_queries;
    ps.nq = nq;
    ps.k = k;
    ps.d_allow = d_allow;
    ps.d_score = d_score;
    ps.d_row = d_row;
    ps.d_count = d_count;
    ps.d_flags = depth == 0 ? d_flags : nullptr;
    ps.st = st;
    ps.seq = seq;
    ps.exact_only = exact_only;
    ps.balance = balance;
    ps.ride = ride;
    ps.big_host_copy = ho && !ride;
    ps.grid = grid;
    ps.G = G;
    ps.nqt = nqt;
    ps.depth = depth;
    ps.sample_rows = sample_rows;
    ps.expected_per_query = ps_expected_per_query;
    ps.b_s = b_s;
    ps.b_r = b_r;
    ps.b_c = b_c;
    if (defer) {
        ps.stats = *acc_stats;
        h->pending = ps;
        return RDX_OK;
    }
    return complete_chunk(h, ps, acc_stats, ho, nullptr);
}

// the host half of a search (see PendingSearch). *redone (if given) = a fallback pass rewrote results after k_finish.
static int complete_chunk_impl(rdx_index* h, const PendingSearch& ps, rdx_search_stats* acc_stats, HostOut* ho, bool* redone) {
    const int64_t nq = ps.nq;
    const int k = ps.k, depth = ps.depth, grid = ps.grid, G = ps.G, nqt = ps.nqt;
    const uint32_t* d_allow = ps.d_allow;
    float* d_score = ps.d_score;
    int64_t* d_row = ps.d_row;
    int32_t* d_count = ps.d_count;
    hipStream_t st = ps.st;
    const bool exact_only = ps.exact_only, balance = ps.balance, ride = ps.ride;
    const int64_t sample_rows = ps.sample_rows;
    const size_t b_s = ps.b_s, b_r = ps.b_r, b_c = ps.b_c;
    const bool prof_all = h->profile == 1 && depth == 0, prof_main = (h->profile == 1 || h->profile == 2) && depth == 0;
    const bool prof_stamps = h->profile == 3 && depth == 0;   // the kernels stamp their own times: nothing extra on the stream
    auto mark = [&](int i) {
        if (prof_all || (prof_main && (i == 3 || i == 4))) (void)hipEventRecord(h->ev[i], st);
    };
    if (redone) *redone = false;
    if (ps.big_host_copy) HIP_TRY(hipStreamSynchronize(st));   // pageable D2H copies: complete only after a stream synchronise
    RDX_TRY(wait_search(h, st, ps.seq));   // the ONE host wait of a search
    const Mailbox& mb = *h->mbox;
    const unsigned long long c_emitted = mb.emitted, c_rescored = mb.rescored;
    const int c_bad = mb.bad;
    if (mb.spec_fail > 0 && depth == 0) h->spec_backoff = 64;   // a speculative threshold was too high: provable thresholds for a while
    // three times the candidates a random corpus would emit: the corpus is clustered — a denser threshold sample for the next searches
    // (search_chunk_impl; re-examined every 256 searches: the denser sample's own emission is what then keeps it on)
    if (depth == 0 && !exact_only && ps.expected_per_query > 0.0) {
        const double per_q = (double)c_emitted / (double)std::max<int64_t>(nq, 1);
        if (per_q > (h->dense_sample > 0 ? 1.5 : 3.0) * ps.expected_per_query) h->dense_sample = 256;
        else if (h->dense_sample > 0) --h->dense_sample;
    }
    // profile = 3: the kernels' own stamps, read NOW — a second-chance pass below runs a nested search whose k_finish overwrites the mailbox
    float stamp_ms_exact = 0.f, stamp_ms_main = 0.f;
    if (prof_stamps) {
        if (exact_only) {
            if (mb.t_last > mb.t_first) stamp_ms_exact = (float)((double)(mb.t_last - mb.t_first) * 1e-5);
        } else if (grid <= 512) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < grid; ++b) {
                if ((b >> 3) >= G * nqt) continue;   // idle workgroups return before they stamp
                t0 = std::min(t0, mb.wg_times[2 * b]);
                t1 = std::max(t1, mb.wg_times[2 * b + 1]);
            }
            if (t1 > t0) stamp_ms_main = (float)((double)(t1 - t0) * 1e-5);
        }
    }
    if (mb.oob) return fail(RDX_ERR_STATE, "internal: the scan computed a corpus address outside the scan copy (RDX_CHECK_BOUNDS build)");
    int n_exact = exact_only ? (int)nq : mb.n_exact;
    if (ride) {
        if (k > 0) {
            std::memcpy(ho->row, h->pin_out, b_r);
            std::memcpy(ho->score, h->pin_out + b_r, b_s);
        }
        std::memcpy(ho->count, h->pin_out + b_r + b_s, b_c);
    }
    if (!exact_only && n_exact > 0 && !c_bad && redone) *redone = true;
    if (!exact_only) {
        if (n_exact > 0 && ho) ho->stale = true;   // a fallback pass rewrites some of the rows copied above
        if (balance) {   // the stamps arrived with the counters: re-weight the XCD shares for the next search
            const unsigned long long* wt = mb.wg_times;
            unsigned long long t0 = ~0ull, tx[8] = {};
            for (int b = 0; b < grid; ++b) {
                if ((b >> 3) >= G * nqt) continue;   // idle workgroups (wpx % nqt != 0) return before they stamp
                t0 = std::min(t0, wt[2 * b]);
                tx[b & 7] = std::max(tx[b & 7], wt[2 * b + 1]);
            }
            double dur[8], mean = 0;
            for (int x = 0; x < 8; ++x) mean += (dur[x] = (double)(tx[x] - t0)) / 8.0;
            if (std::getenv("RDX_DEBUG_XCD")) {   // developer (tools/xcd_spread.py): when each XCD's last (first) workgroup ended, ms after the first start
                std::fprintf(stderr, "xcd end ms:");
                for (int x = 0; x < 8; ++x) {
                    unsigned long long lo = ~0ull;
                    for (int b = x; b < grid; b += 8)
                        if ((b >> 3) < G * nqt) lo = std::min(lo, wt[2 * b + 1]);
                    std::fprintf(stderr, " %.3f(%.3f)", dur[x] * 1e-5, (double)(lo - t0) * 1e-5);
                }
                std::fprintf(stderr, "\n");
            }
            acc_stats->xcd_finish_spread_ms = (float)((*std::max_element(dur, dur + 8) - *std::min_element(dur, dur + 8)) * 1e-5);   // 100 MHz ticks
            acc_stats->xcd_share_min = (float)*std::min_element(h->xw, h->xw + 8);
            acc_stats->xcd_share_max = (float)*std::max_element(h->xw, h->xw + 8);
            if (mean > 0) {
                double sum = 0;
                for (int x = 0; x < 8; ++x) {
                    h->xw[x] *= std::sqrt(mean / std::max(dur[x], 1.0));   // damped: half the correction per search
                    h->xw[x] = std::min(1.5, std::max(0.6, h->xw[x]));
                    sum += h->xw[x];
                }
                for (int x = 0; x < 8; ++x) h->xw[x] *= 8.0 / sum;
            }
        }
        if (n_exact > 0 && !c_bad && depth == 0 && h->retry) {
            // Overflow means "far more rows above the sampled threshold than expected": similar rows stored together
            // (chunks of one document) that the sparse sample missed. Before paying the exact full scan (one fp32 pass over
            // the corpus per 4 queries), give exactly these queries one more MFMA pass as a small, HBM-bound batch with a
            // denser sample and larger segments; what overflows again goes to the exact scan inside that call.
            const int m = n_exact, kk = std::max(k, 1);
            RDX_TRY(h->r_list.ensure((size_t)m * 4));
            RDX_TRY(h->r_q.ensure((size_t)m * h->dim * 4));
            RDX_TRY(h->r_s.ensure((size_t)m * kk * 4));
            RDX_TRY(h->r_r.ensure((size_t)m * kk * 8));
            RDX_TRY(h->r_c.ensure((size_t)m * 4));
            HIP_TRY(hipMemcpyAsync(h->r_list.p, h->exact_list.p, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
            // from the index's own normalised copy (qhat), not from the caller's buffer: an asynchronous caller may have reused
            // that since (include/rdx.h "Lifetimes"); the nested search stores these rows verbatim, so its scores have the same bits
            hipLaunchKernelGGL(k_gather_queries, dim3((m + 3) / 4), dim3(256), 0, st, h->qhat.as<float>(), h->r_list.as<int32_t>(), m, h->dim,
                               h->r_q.as<float>());
            HIP_TRY(hipGetLastError());
            rdx_search_stats sub = {};
            RDX_TRY(search_chunk(h, h->r_q.as<float>(), m, k, d_allow, h->r_s.as<float>(), h->r_r.as<int64_t>(), h->r_c.as<int32_t>(), st,
                                 &sub, 1));
            hipLaunchKernelGGL(k_scatter_topk, dim3(m), dim3(64), 0, st, h->r_s.as<float>(), h->r_r.as<int64_t>(), h->r_c.as<int32_t>(),
                               h->r_list.as<int32_t>(), m, k, d_score, d_row, d_count);
            HIP_TRY(hipGetLastError());
            if (ps.d_flags) HIP_TRY(hipMemsetAsync(ps.d_flags, 0, 16, st));   // the partial is complete now
            HIP_TRY(hipStreamSynchronize(st));   // rdx_search returns with the stream drained
            acc_stats->retried_queries += m;
            acc_stats->emitted += sub.emitted;
            acc_stats->rescored += sub.rescored;
            n_exact = (int)sub.exact_queries;
        } else if (n_exact > 0 && !c_bad) {
            RDX_TRY(run_exact(h, h->exact_list.as<int32_t>(), n_exact, k, d_allow, d_score, d_row, d_count, st));
            if (ps.d_flags) HIP_TRY(hipMemsetAsync(ps.d_flags, 0, 16, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        acc_stats->scan_main_launch_rows = h->rows;
        acc_stats->scan_main_launch_queries = nq;
    }
    mark(6);
    if (c_bad) return fail(RDX_ERR_INVALID, "query embeddings contain NaN or Inf");
    if (prof_all) HIP_TRY(hipEventSynchronize(h->ev[6]));

    acc_stats->sample_rows += exact_only ? 0 : sample_rows;
    acc_stats->emitted += (int64_t)c_emitted;
    acc_stats->rescored += (int64_t)c_rescored;
    acc_stats->exact_queries += n_exact;
    acc_stats->path = exact_only ? 1 : 0;
    if (prof_all) {
        float ms[6] = {};
        for (int i = 0; i < 6; ++i) (void)hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]);
        acc_stats->profiled = 1;
        acc_stats->ms_normalize += ms[0];
        acc_stats->ms_scan_sample += ms[1];
        acc_stats->ms_tau += ms[2];
        acc_stats->ms_scan_main += ms[3];
        acc_stats->ms_refine += ms[4];
        acc_stats->ms_exact += ms[5];
        float tot = 0;
        (void)hipEventElapsedTime(&tot, h->ev[0], h->ev[6]);
        acc_stats->ms_total += tot;
    } else if (prof_main) {
        float ms = 0;
        // non-exact path: both completed (the mailbox came after them in the stream). Exact path: ev[4] sits right behind the
        // kernel whose last block published the mailbox: wait for it (microseconds)
        if (exact_only) (void)hipEventSynchronize(h->ev[4]);
        (void)hipEventElapsedTime(&ms, h->ev[3], h->ev[4]);
        acc_stats->profiled = 2;
        if (exact_only) acc_stats->ms_exact += ms;   // exact path: K5a + K5b
        else acc_stats->ms_scan_main += ms;
    } else if (prof_stamps) {
        // first workgroup start -> last workgroup end of the dominant kernel(s), from the kernels' own 100 MHz stamps
        acc_stats->profiled = 3;
        acc_stats->ms_exact += stamp_ms_exact;
        acc_stats->ms_scan_main += stamp_ms_main;
    }
    return RDX_OK;
}

// run the host half of a search left pending by rdx_search_async (call with h->mu held)
static int finish_pending(rdx_index* h, bool* redone) {
    if (redone) *redone = false;
    if (!h->pending.active) return RDX_OK;
    PendingSearch ps = h->pending;
    h->pending.active = false;
    RDX_TRY(set_device(h));
    rdx_search_stats s = ps.stats;
    const int rc = complete_chunk(h, ps, &s, nullptr, redone);
    h->stats = s;
    return rc;
}

// allow_resident: allow_bits is already device memory whatever `space` says (a resident rdx_mask)
static int search_impl(rdx_index* h, const float* queries, int64_t nq, int k, const uint32_t* allow_bits, bool allow_resident,
                       float* out_score, int64_t* out_row, int32_t* out_count, int space, void* stream) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_search: null index");
    if (nq < 0 || k < 0) return fail(RDX_ERR_INVALID, "rdx_search: nq and k must be >= 0");
    if (k > SELECT_MAX_K) return fail(RDX_ERR_INVALID, "rdx_search: k larger than " + std::to_string(SELECT_MAX_K) + " is not supported");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (nq == 0) return RDX_OK;
    if (!queries || !out_count || (k > 0 && (!out_score || !out_row))) return fail(RDX_ERR_INVALID, "rdx_search: null pointer");
    RDX_TRY(finish_pending(h, nullptr));   // scratch buffers are shared: an asynchronous search completes first
    RDX_TRY(set_device(h));
    // device pointers: the caller's stream as given (NULL = the default stream the caller produced its inputs on)
    hipStream_t st = (space == RDX_DEVICE || stream) ? (hipStream_t)stream : h->own_stream;
    if (h->profile && !h->ev_ok) {
        for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
        h->ev_ok = true;
    }
    rdx_search_stats s = {};
    s.nq = nq;
    s.k = k;
    s.rows = h->rows;

    const uint32_t* d_allow = allow_bits;
    const size_t mask_words = (size_t)((h->rows + 31) / 32);
    if (allow_bits && space == RDX_HOST && !allow_resident && mask_words > 0) {
        // pad to whole 256-row tiles so the scan may read the word of any block it touches
        const size_t pad_words = (size_t)((h->rows + 255) / 256 * 8);
        RDX_TRY(h->mask.ensure(pad_words * 4));
        HIP_TRY(hipMemsetAsync(h->mask.p, 0, pad_words * 4, st));
        HIP_TRY(hipMemcpyAsync(h->mask.p, allow_bits, mask_words * 4, hipMemcpyHostToDevice, st));
        d_allow = h->mask.as<uint32_t>();
    }
    const int64_t CHUNK = 4096;   // queries per pipeline pass (<= 256 * WGs per XCD)
    const int kk = std::max(k, 1);
    for (int64_t q0 = 0; q0 < nq; q0 += CHUNK) {
        const int64_t m = std::min(CHUNK, nq - q0);
        const float* d_q = queries + (size_t)q0 * h->dim;
        float* d_s = out_score ? out_score + (size_t)q0 * k : nullptr;
        int64_t* d_r = out_row ? out_row + (size_t)q0 * k : nullptr;
        int32_t* d_c = out_count + q0;
        if (space == RDX_HOST) {
            RDX_TRY(h->qraw.ensure((size_t)m * h->dim * 4));
            RDX_TRY(h->o_score.ensure((size_t)m * kk * 4));
            RDX_TRY(h->o_row.ensure((size_t)m * kk * 8));
            RDX_TRY(h->o_count.ensure((size_t)m * 4));
            HIP_TRY(hipMemcpyAsync(h->qraw.p, d_q, (size_t)m * h->dim * 4, hipMemcpyHostToDevice, st));
            d_q = h->qraw.as<float>();
            d_s = h->o_score.as<float>();
            d_r = h->o_row.as<int64_t>();
            d_c = h->o_count.as<int32_t>();
        }
        HostOut ho = {out_score ? out_score + (size_t)q0 * k : nullptr, out_row ? out_row + (size_t)q0 * k : nullptr, out_count + q0, false};
        RDX_TRY(search_chunk(h, d_q, m, k, d_allow, d_s, d_r, d_c, st, &s, 0, space == RDX_HOST ? &ho : nullptr));
        if (space == RDX_HOST && ho.stale) {
            if (k > 0) {
                HIP_TRY(hipMemcpyAsync(ho.score, d_s, (size_t)m * k * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(ho.row, d_r, (size_t)m * k * 8, hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipMemcpyAsync(ho.count, d_c, (size_t)m * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    h->stats = s;
    return RDX_OK;
}

extern "C" int rdx_search(rdx_index* h, const float* queries, int64_t nq, int k, const uint32_t* allow_bits, float* out_score,
                          int64_t* out_row, int32_t* out_count, int space, void* stream) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_search: null index");
    std::lock_guard<std::mutex> lk(h->mu);
    return search_impl(h, queries, nq, k, allow_bits, false, out_score, out_row, out_count, space, stream);
}

extern "C" int rdx_search_async(rdx_index* h, const float* queries, int64_t nq, int k, const rdx_mask* mask, float* out_score,
                                int64_t* out_row, int32_t* out_count, int32_t* out_flags, void* stream) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_search_async: null index");
    if (nq < 1 || nq > 4096 || k < 0 || k > SELECT_MAX_K) return fail(RDX_ERR_INVALID, "rdx_search_async: 1 <= nq <= 4096, 0 <= k <= " + std::to_string(SELECT_MAX_K));
    if (!queries || !out_count || (k > 0 && (!out_score || !out_row))) return fail(RDX_ERR_INVALID, "rdx_search_async: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    if (mask && (mask->device != h->device || mask->rows != h->rows))
        return fail(RDX_ERR_STATE, "rdx_search_async: the mask was made for another state of the index");
    RDX_TRY(finish_pending(h, nullptr));
    RDX_TRY(set_device(h));
    if (h->profile && !h->ev_ok) {
        for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
        h->ev_ok = true;
    }
    rdx_search_stats s = {};
    s.nq = nq;
    s.k = k;
    s.rows = h->rows;
    return search_chunk(h, queries, nq, k, mask ? mask->words.as<uint32_t>() : nullptr, out_score, out_row, out_count, (hipStream_t)stream, &s, 0,
                        nullptr, true, out_flags);
}

extern "C" int rdx_search_wait(rdx_index* h, int* redone) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_search_wait: null index");
    std::lock_guard<std::mutex> lk(h->mu);
    bool r = false;
    const int rc = finish_pending(h, &r);
    if (redone) *redone = r ? 1 : 0;
    return rc;
}

extern "C" int rdx_mask_create(rdx_index* h, const uint32_t* allow_bits, int space, rdx_mask** out) {
    if (!h || !allow_bits || !out) return fail(RDX_ERR_INVALID, "rdx_mask_create: null pointer");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    std::lock_guard<std::mutex> lk(h->mu);
    RDX_TRY(finish_pending(h, nullptr));
    RDX_TRY(set_device(h));
    rdx_mask* m = new rdx_mask();
    m->device = h->device;
    m->rows = h->rows;
    const size_t words = (size_t)((h->rows + 31) / 32), pad_words = (size_t)((h->rows + 255) / 256 * 8);
    int rc = m->words.ensure(std::max<size_t>(pad_words, 8) * 4);
    if (rc != RDX_OK) {
        delete m;
        return rc;
    }
    hipStream_t st = h->own_stream;
    hipError_t e = hipMemsetAsync(m->words.p, 0, std::max<size_t>(pad_words, 8) * 4, st);
    if (e == hipSuccess && words > 0) {
        if (space == RDX_DEVICE) e = hipDeviceSynchronize();   // the caller's producer of the bits (any stream) is done
        if (e == hipSuccess)
            e = hipMemcpyAsync(m->words.p, allow_bits, words * 4, space == RDX_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        delete m;
        return fail(RDX_ERR_HIP, std::string("rdx_mask_create: ") + hipGetErrorString(e));
    }
    *out = m;
    return RDX_OK;
}

extern "C" int rdx_mask_destroy(rdx_mask* m) {
    if (!m) return RDX_OK;
    (void)hipSetDevice(m->device);
    delete m;
    return RDX_OK;
}

extern "C" int rdx_search_masked(rdx_index* h, const float* queries, int64_t nq, int k, const rdx_mask* mask, float* out_score,
                                 int64_t* out_row, int32_t* out_count, int space, void* stream) {
    if (!h) return fail(RDX_ERR_INVALID, "rdx_search_masked: null index");
    std::lock_guard<std::mutex> lk(h->mu);
    if (mask && (mask->device != h->device || mask->rows != h->rows))
        return fail(RDX_ERR_STATE, "rdx_search_masked: the mask was made for " + std::to_string(mask->rows) + " rows on device " +
                                       std::to_string(mask->device) + ", the index now holds " + std::to_string(h->rows) +
                                       " (a mask does not outlive a write to the index)");
    return search_impl(h, queries, nq, k, mask ? mask->words.as<uint32_t>() : nullptr, true, out_score, out_row, out_count, space, stream);
}

#ifdef RDX_REFINE_STAMPS
extern "C" int rdx_debug_refine_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(rdx::g_refine_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif
#ifdef RDX_SELECT_STAMPS
extern "C" int rdx_debug_select_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(rdx::g_select_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int rdx_search_last_stats(rdx_index* h, rdx_search_stats* out) {
    if (!h || !out) return fail(RDX_ERR_INVALID, "rdx_search_last_stats: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    *out = h->stats;
    return RDX_OK;
}

extern "C" int rdx_merge_topk(int device, const float* part_score, const int64_t* part_row, const int32_t* part_count, int n_parts,
                              int64_t nq, int k, float* out_score, int64_t* out_row, int32_t* out_count, int space, void* stream) {
    if (n_parts < 1 || n_parts > 64 || nq < 0 || k < 0 || k > SELECT_MAX_K) return fail(RDX_ERR_INVALID, "rdx_merge_topk: bad shape");
    if (space != RDX_HOST && space != RDX_DEVICE) return fail(RDX_ERR_INVALID, "space must be RDX_HOST or RDX_DEVICE");
    if (nq == 0) return RDX_OK;
    if (!part_count || !out_count || (k > 0 && (!part_score || !part_row || !out_score || !out_row)))
        return fail(RDX_ERR_INVALID, "rdx_merge_topk: null pointer");
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t np = (size_t)n_parts * nq;
    const int kk = std::max(k, 1);
    DevBuf ps, pr, pc, os, orow, oc;
    const float* d_ps = part_score;
    const int64_t* d_pr = part_row;
    const int32_t* d_pc = part_count;
    float* d_os = out_score;
    int64_t* d_or = out_row;
    int32_t* d_oc = out_count;
    if (space == RDX_HOST) {
        RDX_TRY(ps.ensure(np * kk * 4));
        RDX_TRY(pr.ensure(np * kk * 8));
        RDX_TRY(pc.ensure(np * 4));
        RDX_TRY(os.ensure((size_t)nq * kk * 4));
        RDX_TRY(orow.ensure((size_t)nq * kk * 8));
        RDX_TRY(oc.ensure((size_t)nq * 4));
        if (k > 0) {
            HIP_TRY(hipMemcpyAsync(ps.p, part_score, np * k * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(pr.p, part_row, np * k * 8, hipMemcpyHostToDevice, st));
        }
        HIP_TRY(hipMemcpyAsync(pc.p, part_count, np * 4, hipMemcpyHostToDevice, st));
        d_ps = ps.as<float>();
        d_pr = pr.as<int64_t>();
        d_pc = pc.as<int32_t>();
        d_os = os.as<float>();
        d_or = orow.as<int64_t>();
        d_oc = oc.as<int32_t>();
    }
    DevBuf t_s[2], t_r[2], t_c[2];   // the fold's intermediate lists (large k only)
    if ((int64_t)n_parts * k <= MERGE_MAX) {
        hipLaunchKernelGGL(k_merge, dim3((int)nq), dim3(256), 0, st, d_ps, d_pr, d_pc, (int64_t)nq * k, (int64_t)nq * k, (int64_t)nq, n_parts, nq, k,
                           d_os, d_or, d_oc);
        HIP_TRY(hipGetLastError());
    } else {
        // more candidates per query than k_merge ranks in LDS (several shards at k > 2048 / n_parts): fold the parts pairwise
        const size_t nk = (size_t)nq * k;
        for (int j = 0; j < 2 && n_parts > 2; ++j) {
            RDX_TRY(t_s[j].ensure(nk * 4));
            RDX_TRY(t_r[j].ensure(nk * 8));
            RDX_TRY(t_c[j].ensure((size_t)nq * 4));
        }
        const float* a_s = d_ps;
        const int64_t* a_r = d_pr;
        const int32_t* a_c = d_pc;
        for (int p = 1; p < n_parts; ++p) {
            const bool last = p == n_parts - 1;
            float* o_s = last ? d_os : t_s[p & 1].as<float>();
            int64_t* o_r = last ? d_or : t_r[p & 1].as<int64_t>();
            int32_t* o_c = last ? d_oc : t_c[p & 1].as<int32_t>();
            hipLaunchKernelGGL(k_merge_pair, dim3((int)nq), dim3(256), 0, st, a_s, a_r, a_c, d_ps + (size_t)p * nk, d_pr + (size_t)p * nk,
                               d_pc + (size_t)p * nq, k, o_s, o_r, o_c);
            HIP_TRY(hipGetLastError());
            a_s = o_s;
            a_r = o_r;
            a_c = o_c;
        }
        if (space == RDX_DEVICE && n_parts > 2) HIP_TRY(hipStreamSynchronize(st));   // the intermediates are freed on return
    }
    if (space == RDX_HOST) {
        if (k > 0) {
            HIP_TRY(hipMemcpyAsync(out_score, d_os, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_row, d_or, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipMemcpyAsync(out_count, d_oc, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (DevBuf* b : {&ps, &pr, &pc, &os, &orow, &oc}) b->release();
    }
    return RDX_OK;
}

extern "C" int rdx_signal_create(int device, rdx_signal** out) {
    if (!out) return fail(RDX_ERR_INVALID, "rdx_signal_create: null out pointer");
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(p, 0, 64);
    void* d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, p, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(p);
        return fail(RDX_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    rdx_signal* s = new rdx_signal();
    s->device = device;
    s->host = reinterpret_cast<unsigned long long*>(p);
    s->dev = reinterpret_cast<unsigned long long*>(d);
    *out = s;
    return RDX_OK;
}

extern "C" int rdx_signal_destroy(rdx_signal* s) {
    if (!s) return RDX_OK;
    (void)hipSetDevice(s->device);
    if (s->host) (void)hipHostFree(s->host);
    delete s;
    return RDX_OK;
}

extern "C" int rdx_signal_wait(rdx_signal* s, void* stream, int32_t* value) {
    if (!s || !value) return fail(RDX_ERR_INVALID, "rdx_signal_wait: null pointer");
    if (s->seq == 0) return fail(RDX_ERR_STATE, "rdx_signal_wait: no merge has been given this signal");
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long w = 0;
    const int rc = wait_word([&] { return ((w = __atomic_load_n(s->host, __ATOMIC_ACQUIRE)) >> 1) == s->seq; }, (hipStream_t)stream);
    if (rc == 1) return fail(RDX_ERR_HIP, "internal: the merge completed without publishing its signal");
    if (rc != 0) return rc;
    *value = (int32_t)(w & 1ull);
    return RDX_OK;
}

// the packed layout one rank contributes to the all-gather: rows i64[nq][k] | scores f32[nq][k] | counts i32[nq] | flags i32[4]
extern "C" int rdx_merge_topk_packed(int device, const void* packed, int64_t part_stride, int n_parts, int64_t nq, int k,
                                     float* out_score, int64_t* out_row, int32_t* out_count, rdx_signal* sig, void* stream) {
    if (n_parts < 1 || n_parts > 64 || nq < 0 || k < 1) return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: bad shape");
    if ((int64_t)n_parts * k > MERGE_MAX) return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: n_parts * k exceeds " + std::to_string(MERGE_MAX));
    const int64_t flags_off = nq * k * 12 + nq * 4;
    if (part_stride % 16 != 0 || part_stride < flags_off + 4 * RDX_PACKED_FLAGS)
        return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: part_stride must be a multiple of 16 covering one packed partial (rows | scores | counts | flags)");
    if (sig && sig->device != device) return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: the signal belongs to another device");
    if (nq == 0) {
        if (sig) return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: a signal needs nq >= 1");
        return RDX_OK;
    }
    if (!packed || !out_score || !out_row || !out_count) return fail(RDX_ERR_INVALID, "rdx_merge_topk_packed: null pointer");
    HIP_TRY(hipSetDevice(device));
    const char* b = reinterpret_cast<const char*>(packed);
    const unsigned long long seq = sig ? ++sig->seq : 0;
    hipLaunchKernelGGL(k_merge, dim3((int)nq), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(b + nq * k * 8),
                       reinterpret_cast<const int64_t*>(b), reinterpret_cast<const int32_t*>(b + nq * k * 12), part_stride / 4,
                       part_stride / 8, part_stride / 4, n_parts, nq, k, out_score, out_row, out_count,
                       reinterpret_cast<const int32_t*>(b + flags_off), part_stride / 4, sig ? sig->dev : (unsigned long long*)nullptr, seq);
    HIP_TRY(hipGetLastError());
    return RDX_OK;
}
