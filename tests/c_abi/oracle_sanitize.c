/* Sanitizer driver for the CPU oracle (test infrastructure): compiled TOGETHER with oracle/rdx_oracle.c under
 * -fsanitize=address,undefined by tests/test_sanitizers.py and run on seeded inputs that reach every loop bound the oracle has:
 * k larger than the corpus, k larger than the allowed rows, an empty corpus, an all-masked corpus, duplicate rows (the tie
 * rule's comparator), a zero row (the 1e-12 floor), a merge whose parts are short or empty. Prints a checksum of everything it
 * computed; the test compares it with the checksum of the un-instrumented library on the same inputs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void rdxo_normalize_rows(const float* in, int64_t n, int d, float* out);
float rdxo_score(const float* qhat, const float* chat, int d);
void rdxo_scores(const float* corpus_hat, int64_t N, int d, const float* qhat, float* out);
void rdxo_cosine_topk(const float* corpus_hat, int64_t N, int d, const float* q_raw, int64_t B, int k, const uint32_t* allow,
                      float* out_score, int64_t* out_row, int32_t* out_count);
void rdxo_merge_topk(const float* part_score, const int64_t* part_row, const int32_t* part_count, int P, int64_t B, int k,
                     float* out_score, int64_t* out_row, int32_t* out_count);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand(void) { /* xorshift64*, mapped to [-1, 1) */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (float)((double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 0x100000001B3ull;
    return h;
}

int main(void) {
    uint64_t sum = 0xCBF29CE484222325ull;
    const int dims[] = {4, 64, 260, 1024};
    for (int di = 0; di < 4; ++di) {
        const int d = dims[di];
        const int64_t N = 257;
        float* raw = (float*)malloc((size_t)N * d * sizeof(float));
        float* hat = (float*)malloc((size_t)N * d * sizeof(float));
        for (int64_t i = 0; i < N * d; ++i) raw[i] = frand();
        memcpy(raw + 5 * d, raw + 3 * d, (size_t)d * sizeof(float));        /* duplicate rows: tie rule */
        memset(raw + 7 * d, 0, (size_t)d * sizeof(float));                   /* a zero row: the 1e-12 floor */
        rdxo_normalize_rows(raw, N, d, hat);
        sum = fnv(sum, hat, (size_t)N * d * sizeof(float));
        const int64_t B = 3;
        float* q = (float*)malloc((size_t)B * d * sizeof(float));
        for (int64_t i = 0; i < B * d; ++i) q[i] = frand();
        memcpy(q, raw + 3 * d, (size_t)d * sizeof(float));                   /* a stored row as query */
        uint32_t allow[9];
        const int ks[] = {1, 10, 257, 300};
        for (int ki = 0; ki < 4; ++ki) {
            const int k = ks[ki];
            float* s = (float*)malloc((size_t)B * k * sizeof(float));
            int64_t* r = (int64_t*)malloc((size_t)B * k * sizeof(int64_t));
            int32_t c[3];
            for (int mode = 0; mode < 4; ++mode) {                           /* no mask, sparse mask, all masked, one row */
                for (int w = 0; w < 9; ++w) allow[w] = mode == 1 ? 0x11111111u : (mode == 3 && w == 8 ? 1u : 0u);
                rdxo_cosine_topk(hat, N, d, q, B, k, mode ? allow : NULL, s, r, c);
                sum = fnv(sum, s, (size_t)B * k * sizeof(float));
                sum = fnv(sum, r, (size_t)B * k * sizeof(int64_t));
                sum = fnv(sum, c, sizeof(c));
            }
            rdxo_cosine_topk(hat, 0, d, q, B, k, NULL, s, r, c);             /* empty corpus */
            sum = fnv(sum, c, sizeof(c));
            /* merge: three parts = the same search over three row ranges (global ids), one of them empty */
            float* ps = (float*)malloc((size_t)3 * B * k * sizeof(float));
            int64_t* pr = (int64_t*)malloc((size_t)3 * B * k * sizeof(int64_t));
            int32_t pc[9];
            const int64_t lo[4] = {0, 100, 100, N};
            for (int p = 0; p < 3; ++p) {
                rdxo_cosine_topk(hat + lo[p] * d, lo[p + 1] - lo[p], d, q, B, k, NULL, ps + (size_t)p * B * k, pr + (size_t)p * B * k, pc + p * B);
                for (int64_t i = 0; i < B * k; ++i)
                    if (pr[(size_t)p * B * k + i] >= 0) pr[(size_t)p * B * k + i] += lo[p];
            }
            float* ms = (float*)malloc((size_t)B * k * sizeof(float));
            int64_t* mr = (int64_t*)malloc((size_t)B * k * sizeof(int64_t));
            int32_t mc[3];
            rdxo_merge_topk(ps, pr, pc, 3, B, k, ms, mr, mc);
            rdxo_cosine_topk(hat, N, d, q, B, k, NULL, s, r, c);
            if (memcmp(ms, s, (size_t)B * k * sizeof(float)) || memcmp(mr, r, (size_t)B * k * sizeof(int64_t)) || memcmp(mc, c, sizeof(c))) {
                fprintf(stderr, "merge of three ranges != whole search (d %d, k %d)\n", d, k);
                return 2;
            }
            sum = fnv(sum, ms, (size_t)B * k * sizeof(float));
            free(ms); free(mr); free(ps); free(pr); free(s); free(r);
        }
        float* all = (float*)malloc((size_t)N * sizeof(float));
        float* qh = (float*)malloc((size_t)d * sizeof(float));
        rdxo_normalize_rows(q, 1, d, qh);
        rdxo_scores(hat, N, d, qh, all);
        if (all[3] != rdxo_score(qh, hat + 3 * d, d) || fabsf(all[3] - 1.0f) > 1e-6f) return 3;
        sum = fnv(sum, all, (size_t)N * sizeof(float));
        free(all); free(qh); free(q); free(raw); free(hat);
    }
    printf("oracle sanitize checksum %016llx\n", (unsigned long long)sum);
    return 0;
}
