/*
 * rdx_oracle.c — CPU restatement of the reference's dense-retrieval hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library.
 * The product path (rag_dpo_amd/) never does; it fails loudly when the HIP library is missing.
 *
 * PARITY UNPINNED for the top-k arithmetic itself: in the reference that arithmetic lives in the
 * third-party wheel chromadb==1.4.1 (reference requirements.txt:32), which is neither vendored under
 * /root/reference nor installed here, and the reference holds no golden ids/distances for it
 * (SURVEY.md §4, §8c). What is restated is Chroma's published contract for a collection created with
 * {"hnsw:space": "cosine"} (reference src/processing/create_chromadb_index.py:100-106):
 *      distance(q, c) = 1 - <q, c> / (|q| |c|),   results by ascending distance,
 * evaluated EXACTLY (brute force) instead of through HNSW, and anchored on the reference's call sites:
 *   - embeddings are L2-normalised with x / max(|x|_2, 1e-12)
 *     (SentenceTransformer.encode(normalize_embeddings=True), reference src/utils/embedding_provider.py:139-145)
 *   - query(query_embeddings, n_results, where) -> ids/distances ascending
 *     (reference src/rag/retriever.py:215-220, 380-385, 472-494)
 *   - `where` is a PRE-filter: filtered queries still return n_results hits when enough rows pass
 *     (reference tasks/lessons.md:53-57)
 * The tie rule (equal score -> lower insertion index first) is this build's; Chroma documents none.
 * The oracle is pinned only by known-answer tests (tests/test_oracle.py) and by an independent numpy
 * float64 restatement (oracle/oracle.py: topk_numpy).
 *
 * Arithmetic (shared with the HIP kernels so that ids can be compared bit-exactly):
 *   "lane order" sum over a d-vector (d % 4 == 0): the vector is cut into float4 groups g = 0..d/4-1;
 *   virtual lane l = g % 64 accumulates its groups in increasing g, elements in increasing index, in
 *   fp64 (every fp32*fp32 product is exact in fp64, so each step rounds once); the 64 lane sums are
 *   then combined by the butterfly p[l] += p[l ^ m] for m = 32,16,8,4,2,1.
 *   normalise: n2 = lane-order sum of x*x; den = max(sqrt(n2), 1e-12); y[i] = (float)((double)x[i] / den)
 *   score:     s = (float) lane-order sum of qhat[i]*chat[i]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 64

static double lane_order_dot(const float* a, const float* b, int d) {
    double p[LANES];
    for (int l = 0; l < LANES; ++l) p[l] = 0.0;
    const int n4 = d / 4;
    for (int g0 = 0; g0 < n4; g0 += LANES) {
        const int lim = (n4 - g0) < LANES ? (n4 - g0) : LANES;
        for (int l = 0; l < lim; ++l) {
            const float* pa = a + 4 * (g0 + l);
            const float* pb = b + 4 * (g0 + l);
            double acc = p[l];
            acc += (double)pa[0] * (double)pb[0];
            acc += (double)pa[1] * (double)pb[1];
            acc += (double)pa[2] * (double)pb[2];
            acc += (double)pa[3] * (double)pb[3];
            p[l] = acc;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        double t[LANES];
        for (int l = 0; l < LANES; ++l) t[l] = p[l] + p[l ^ m];
        memcpy(p, t, sizeof(p));
    }
    return p[0];
}

/* out[i] = in[i] / max(|in|, 1e-12), reference src/utils/embedding_provider.py:144 */
void rdxo_normalize_rows(const float* in, int64_t n, int d, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const float* x = in + r * (int64_t)d;
        float* y = out + r * (int64_t)d;
        const double n2 = lane_order_dot(x, x, d);
        double den = sqrt(n2);
        if (den < 1e-12) den = 1e-12;
        for (int i = 0; i < d; ++i) y[i] = (float)((double)x[i] / den);
    }
}

float rdxo_score(const float* qhat, const float* chat, int d) {
    return (float)lane_order_dot(qhat, chat, d);
}

/* all scores of one normalised query against a normalised corpus */
void rdxo_scores(const float* corpus_hat, int64_t N, int d, const float* qhat, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N; ++r) out[r] = (float)lane_order_dot(qhat, corpus_hat + r * (int64_t)d, d);
}

typedef struct {
    float s;
    int64_t r;
} ent;

/* a ranks before b: higher score, then lower row id */
static inline int before(ent a, ent b) { return a.s > b.s || (a.s == b.s && a.r < b.r); }

static void sift_down(ent* h, int n, int i) { /* heap root = the WORST kept entry */
    for (;;) {
        int w = i, l = 2 * i + 1, r = l + 1;
        if (l < n && before(h[w], h[l])) w = l;
        if (r < n && before(h[w], h[r])) w = r;
        if (w == i) return;
        ent t = h[i];
        h[i] = h[w];
        h[w] = t;
        i = w;
    }
}

static int cmp_ent(const void* pa, const void* pb) {
    ent a = *(const ent*)pa, b = *(const ent*)pb;
    return before(a, b) ? -1 : (before(b, a) ? 1 : 0);
}

/*
 * collection.query restated (reference src/rag/retriever.py:215-220): brute-force cosine top-k.
 *   corpus_hat [N][d] rows already normalised by rdxo_normalize_rows (what collection.add stored)
 *   q_raw      [B][d] raw query embeddings (normalised here)
 *   allow      NULL or ceil(N/32) words, bit set = row passes the `where` pre-filter
 *   out_score/out_row [B][k], out_count[B]; unused tail = (-inf, -1)
 */
void rdxo_cosine_topk(const float* corpus_hat, int64_t N, int d, const float* q_raw, int64_t B, int k,
                      const uint32_t* allow, float* out_score, int64_t* out_row, int32_t* out_count) {
    float* qhat = (float*)malloc((size_t)B * d * sizeof(float));
    rdxo_normalize_rows(q_raw, B, d, qhat);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        ent* heap = (ent*)malloc((size_t)(k > 0 ? k : 1) * sizeof(ent));
        int n = 0;
        const float* q = qhat + b * (int64_t)d;
        for (int64_t r = 0; r < N; ++r) {
            if (allow && !((allow[r >> 5] >> (r & 31)) & 1u)) continue;
            ent e;
            e.s = (float)lane_order_dot(q, corpus_hat + r * (int64_t)d, d);
            e.r = r;
            if (n < k) {
                heap[n++] = e;
                if (n == k)
                    for (int i = n / 2 - 1; i >= 0; --i) sift_down(heap, n, i);
            } else if (k > 0 && before(e, heap[0])) {
                heap[0] = e;
                sift_down(heap, n, 0);
            }
        }
        qsort(heap, (size_t)n, sizeof(ent), cmp_ent);
        for (int i = 0; i < k; ++i) {
            out_score[b * (int64_t)k + i] = i < n ? heap[i].s : -INFINITY;
            out_row[b * (int64_t)k + i] = i < n ? heap[i].r : -1;
        }
        out_count[b] = n;
        free(heap);
    }
    free(qhat);
}

/* merge per-shard partials [P][B][k] (global row ids) into the global top-k */
void rdxo_merge_topk(const float* part_score, const int64_t* part_row, const int32_t* part_count, int P,
                     int64_t B, int k, float* out_score, int64_t* out_row, int32_t* out_count) {
    for (int64_t b = 0; b < B; ++b) {
        ent* all = (ent*)malloc((size_t)P * (k > 0 ? k : 1) * sizeof(ent));
        int n = 0;
        for (int p = 0; p < P; ++p) {
            const int c = part_count[(int64_t)p * B + b];
            for (int i = 0; i < c; ++i) {
                all[n].s = part_score[((int64_t)p * B + b) * k + i];
                all[n].r = part_row[((int64_t)p * B + b) * k + i];
                ++n;
            }
        }
        qsort(all, (size_t)n, sizeof(ent), cmp_ent);
        for (int i = 0; i < k; ++i) {
            out_score[b * (int64_t)k + i] = i < n ? all[i].s : -INFINITY;
            out_row[b * (int64_t)k + i] = i < n ? all[i].r : -1;
        }
        out_count[b] = n < k ? n : k;
        free(all);
    }
}

int rdxo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
