/* C-ABI smoke test: a plain C program (gcc, no HIP headers, no Python) drives librdx through include/rdx.h the way a
 * host-language binding would. Built and run by tests/test_gpu_cabi.py on the GPU box. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rdx.h"

#define CHECK(x)                                                            \
    do {                                                                    \
        int rc_ = (x);                                                      \
        if (rc_) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rdx_last_error());   \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    const int dim = 1024, n = 3000, nq = 2, k = 5;
    float* rows = (float*)calloc((size_t)n * dim, sizeof(float));
    float* q = (float*)calloc((size_t)nq * dim, sizeof(float));
    for (int i = 0; i < n; ++i) {           /* row i = e_(i mod dim) scaled by (i+1): direction only matters */
        rows[(size_t)i * dim + (i % dim)] = (float)(i + 1);
        rows[(size_t)i * dim + ((i + 1) % dim)] = (i >= dim) ? 0.5f * (float)(i + 1) : 0.f;
    }
    q[7] = 1.f;                              /* query 0 = e_7 */
    q[dim + 9] = -2.f;                       /* query 1 = -e_9 */
    rdx_index* h = NULL;
    CHECK(rdx_index_create(0, dim, &h));
    CHECK(rdx_index_add(h, rows, n, RDX_HOST));
    int64_t cnt = 0;
    CHECK(rdx_index_count(h, &cnt));
    if (cnt != n) return 2;
    float score[2 * 5];
    int64_t row[2 * 5];
    int32_t got[2];
    CHECK(rdx_search(h, q, nq, k, NULL, score, row, got, RDX_HOST, NULL));
    /* rows 7 (pure e_7, cos 1) then 1031 and 2055 (e_7 + 0.5 e_8, cos 2/sqrt5), then 1030 / 2054 (e_6 + .5 e_7, cos 1/sqrt5) */
    if (got[0] != k || row[0] != 7 || row[1] != 1031 || row[2] != 2055 || row[3] != 1030 || row[4] != 2054) {
        fprintf(stderr, "unexpected ids %ld %ld %ld %ld %ld\n", (long)row[0], (long)row[1], (long)row[2], (long)row[3], (long)row[4]);
        return 3;
    }
    if (fabsf(score[0] - 1.f) > 1e-6f || fabsf(score[1] - 0.8944272f) > 1e-6f || score[1] != score[2]) return 4;
    if (!(score[5] <= 0.f)) return 5;        /* nothing points along -e_9 except with negative or zero cosine */
    int rc = rdx_index_add(h, rows, -1, RDX_HOST);
    if (rc != RDX_ERR_INVALID || rdx_last_error()[0] == 0) return 6;   /* error convention: code + message */
    CHECK(rdx_index_destroy(h));
    printf("c-abi smoke ok: %s\n", "ids and scores as expected");
    free(rows);
    free(q);
    return 0;
}
