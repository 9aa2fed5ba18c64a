/* Error paths of the C-ABI that need no GPU (argument checks come before any HIP call): a plain C program built with
 * -fsanitize=address,undefined by tests/test_sanitizers.py. Every call must return its error code with a message from
 * rdx_last_error(), touch none of the (deliberately tiny or NULL) buffers, and leave the process alive. On a box without a GPU
 * rdx_index_create itself fails cleanly (RDX_ERR_HIP); on a GPU box (tests/test_gpu_cabi.py) the index calls below run against a
 * real handle. */
#include <stdio.h>
#include <string.h>

#include "rdx.h"

#define EXPECT(code, call)                                                                   \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (!(rc_ == (code) || (no_device && rc_ == RDX_ERR_HIP)) || rdx_last_error() == NULL || rdx_last_error()[0] == 0) { \
            fprintf(stderr, "%s -> %d (wanted %d): %s\n", #call, rc_, (code), rdx_last_error()); \
            return 1;                                                                        \
        }                                                                                    \
        ++n_checked;                                                                         \
    } while (0)

int main(void) {
    int n_checked = 0, no_device = 0, n_dev = 0;
    if (rdx_version() != RDX_ABI_VERSION) return 2;
    /* without a GPU the few checks that sit behind a call's hipSetDevice answer RDX_ERR_HIP instead: accepted there only */
    if (rdx_device_count(&n_dev) != RDX_OK || n_dev < 1) no_device = 1;
    float one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    float out[8];
    int32_t tf[4] = {0, 0, 0, 0};
    int64_t rows[4] = {0, 1, 2, 3};
    _Alignas(16) unsigned short h16[64];
    memset(h16, 0, sizeof(h16));
    rdx_index* h = NULL;
    EXPECT(RDX_ERR_INVALID, rdx_device_count(NULL));
    EXPECT(RDX_ERR_INVALID, rdx_index_create(0, 1024, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_index_create(0, 0, &h));
    EXPECT(RDX_ERR_INVALID, rdx_index_create(0, 1023, &h));
    EXPECT(RDX_ERR_INVALID, rdx_index_create(0, 1 << 20, &h));
    EXPECT(RDX_ERR_INVALID, rdx_set_wait_policy(-1, 0));
    EXPECT(RDX_ERR_INVALID, rdx_set_wait_policy(0, -5));
    if (rdx_set_wait_policy(400, 0) != RDX_OK) return 3;
    EXPECT(RDX_ERR_INVALID, rdx_l2_normalize(0, one, -1, 8, out, RDX_HOST, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_l2_normalize(0, one, 1, 6, out, RDX_HOST, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_l2_normalize(0, one, 1, 8, out, 7, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_l2_normalize(99, one, 1, 8, out, RDX_HOST, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_attention_f16(0, h16, tf, tf, 4, 1, 32, 0.125f, 0, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_attention_f16(0, NULL, tf, tf, 4, 1, 64, 0.125f, 0, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_attention_f16(0, (char*)h16 + 2, tf, tf, 4, 1, 64, 0.125f, 0, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_attention_small_f16(0, h16, tf, 33, 1, 64, 0.125f, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_attention_small_f16(0, h16, NULL, 4, 1, 64, 0.125f, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_add_layernorm_f16(0, h16, h16, h16, h16, 1e-5f, 1, 768, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_layernorm_rows_f16(0, h16, h16, h16, 1e-5f, 1, 100, out, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_embed_f16(0, rows, rows, h16, h16, h16, 4, 768, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_linear_small_f16(0, h16, h16, h16, 300, 16, 512, 0, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_linear_small_f16(0, h16, h16, h16, 4, 24, 512, 0, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_stage_f16(0, h16, NULL, NULL, NULL, 0.f, NULL, h16, h16, NULL, 33, 16, 512, 0, 16, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_stage_f16(0, h16, NULL, NULL, NULL, 0.f, NULL, h16, h16, NULL, 4, 16, 768, 0, 16, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_stage_f16(0, h16, NULL, NULL, NULL, 0.f, NULL, h16, h16, NULL, 4, 16, 512, 2, 16, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_stage_f16(0, h16, NULL, NULL, NULL, 0.f, NULL, h16, h16, NULL, 4, 16, 512, 0, 5, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_enc_stage_f16(0, h16, rows, h16, h16, 1e-5f, NULL, h16, h16, NULL, 4, 16, 512, 0, 16, h16, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_merge_topk(0, out, rows, tf, 0, 1, 1, out, rows, tf, RDX_HOST, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_merge_topk(0, out, rows, tf, 1, 1, -1, out, rows, tf, RDX_HOST, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_merge_topk_packed(0, h16, 8, 1, 1, 1, out, rows, tf, NULL, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_merge_topk_packed(0, h16, 32, 64, 1, 100, out, rows, tf, NULL, NULL));
    EXPECT(RDX_ERR_INVALID, rdx_signal_create(0, NULL));
    int rc = rdx_index_create(0, 8, &h);
    if (rc == RDX_OK) { /* a GPU is present: the handle-level checks */
        EXPECT(RDX_ERR_INVALID, rdx_index_add(h, one, -1, RDX_HOST));
        EXPECT(RDX_ERR_INVALID, rdx_index_add(h, one, 1, 9));
        EXPECT(RDX_ERR_INVALID, rdx_index_set_option(h, "no_such_option", 1));
        EXPECT(RDX_ERR_INVALID, rdx_search(h, one, 1, -1, NULL, out, rows, tf, RDX_HOST, NULL));
        EXPECT(RDX_ERR_INVALID, rdx_index_get(h, NULL, 1, out, RDX_HOST));
        if (rdx_index_destroy(h) != RDX_OK) return 4;
    } else if (rc != RDX_ERR_HIP && rc != RDX_ERR_INVALID) {
        fprintf(stderr, "rdx_index_create without a GPU -> %d: %s\n", rc, rdx_last_error());
        return 5;
    }
    printf("c-abi error paths ok: %d checks%s\n", n_checked, rc == RDX_OK ? " (with a device)" : " (no device)");
    return 0;
}
