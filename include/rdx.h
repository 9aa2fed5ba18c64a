/*
 * rdx.h — C-ABI of librdx, the MI355X (gfx950) dense-retrieval hot path that stands in for
 * RAG-DPO's `embedding_provider.embed(...)` L2-normalise step and Chroma `collection.query(...)`.
 *
 * The reference (MatJoss/RAG-DPO) has no FFI of its own: its boundary is two duck-typed Python
 * objects. Each entry point below names the reference call it replaces; the
 * Python mirror of those objects (rag_dpo_amd/collection.py, embedding_provider.py) binds these
 * symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 (RDX_OK) or an RDX_ERR_* code; rdx_last_error() gives the
 *     thread-local message of the last failure on the calling thread.
 *   - `space` says where EVERY pointer argument of that call lives: RDX_HOST or RDX_DEVICE
 *     (device = the index's HIP device). Caller owns all buffers.
 *   - `stream` is a hipStream_t passed as void*. RDX_DEVICE calls run on exactly that stream (NULL =
 *     HIP's default stream, as with any HIP API), i.e. ordered after whatever the caller enqueued
 *     there to produce the inputs; rdx_search additionally synchronises the stream once before it
 *     returns (overflow check). RDX_HOST calls are synchronous; NULL = the library's own stream.
 *     Calls WITHOUT a stream argument (add / update / get / compact) given RDX_DEVICE pointers first
 *     wait for all work enqueued on the device so far (hipDeviceSynchronize) and are complete on return.
 *   - rows are addressed by their insertion index ("row id", int64, 0-based); the Python layer
 *     maps row ids to Chroma string ids / documents / metadatas.
 *   - scores are cosine similarities in fp32: exact dot product of the two L2-normalised fp32
 *     vectors accumulated in fp64 in a fixed order (oracle/rdx_oracle.c states the order), rounded
 *     once to fp32. Chroma's `distance` is `1.0f - score` ("hnsw:space": "cosine",
 *     reference src/processing/create_chromadb_index.py:100-106).
 *   - result order per query: score descending, ties by ascending row id (= ascending distance,
 *     the order `RAGRetriever._parse_chromadb_results` consumes, reference src/rag/retriever.py:472-494).
 */
#ifndef RDX_H
#define RDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDX_OK 0
#define RDX_ERR_INVALID 1 /* bad argument: shape, k, NaN/Inf in input, unknown option */
#define RDX_ERR_HIP 2     /* HIP runtime failure (message carries hipGetErrorString) */
#define RDX_ERR_NOMEM 3   /* device or host allocation failed */
#define RDX_ERR_STATE 4   /* call not valid in the index's current state */

#define RDX_HOST 0
#define RDX_DEVICE 1

#define RDX_ABI_VERSION 3 /* 2: flags word in the packed partial, rdx_signal, rdx_search_async(out_flags); 3: the single-question encoder stages */

typedef struct rdx_index rdx_index; /* opaque: one corpus shard resident in one GPU's HBM */

/* Library / device ---------------------------------------------------------------------------- */
int rdx_version(void);
const char* rdx_last_error(void);
int rdx_device_count(int* n);
/* How the host waits for a search (rdx_search, rdx_search_wait, rdx_signal_wait), process-wide. Default (400, 0): spin on the
 * search's pinned mailbox word for spin_us, then poll it with sched_yield() between looks — lowest latency, one host core busy per
 * waiting thread for the length of the search. sleep_us > 0: after the spin, sleep that long between looks (a server whose
 * sessions share the cores — the reference serves concurrent Streamlit sessions from one process, app.py:42-43 — wants e.g.
 * (50, 100): small searches still end inside the spin, a 15 ms scan costs ~1 % of a core and is noticed <= 100 us late).
 * Environment RDX_WAIT_SPIN_US / RDX_WAIT_SLEEP_US set the initial values. Speed only. */
int rdx_set_wait_policy(int spin_us, int sleep_us);

/* Index lifecycle — replaces chromadb `create_collection(..., {"hnsw:space": "cosine"})` /
 * `get_collection` (reference create_chromadb_index.py:100-106,112; app.py:58-59). */
int rdx_index_create(int device, int dim, rdx_index** out);
int rdx_index_destroy(rdx_index* h);
int rdx_index_dim(const rdx_index* h, int* dim);
/* `collection.count()` (reference src/rag/bm25_index.py:200, app.py:108). */
int rdx_index_count(const rdx_index* h, int64_t* rows);
int rdx_index_reserve(rdx_index* h, int64_t rows);

/* `collection.add(embeddings=...)` (reference create_chromadb_index.py:374-379,
 * ingest_enterprise.py:241-246): append n raw fp32 rows [n][dim]; each is L2-normalised on the
 * device (K1) into the fp32 master copy and the tiled fp16 scan copy. Row ids continue from
 * count(). Rows containing NaN/Inf are rejected (RDX_ERR_INVALID) and nothing is added. */
int rdx_index_add(rdx_index* h, const float* rows, int64_t n, int space);
/* Config 5 (BASELINE.json): corpus delivered as bf16 (raw 16-bit patterns); the master copy
 * holds the bf16 values widened to fp32 and normalised like any other row. */
int rdx_index_add_bf16(rdx_index* h, const uint16_t* rows, int64_t n, int space);
/* Reload of a persisted collection (chromadb's PersistentClient returns stored vectors unchanged, reference app.py:58-59
 * reads what create_chromadb_index.py wrote): rows are values previously returned by rdx_index_get, i.e. already
 * L2-normalised; they are stored VERBATIM (no second normalisation, which could move a component by an ulp), so scores
 * after a reload are bit-identical to those before it. NaN/Inf rows are rejected as in rdx_index_add. */
int rdx_index_add_stored(rdx_index* h, const float* rows, int64_t n, int space);
/* `collection.update(ids=, embeddings=)` / upsert: overwrite existing rows in place. */
int rdx_index_update(rdx_index* h, const int64_t* row_ids, const float* rows, int64_t n, int space);
/* `collection.get(include=["embeddings"])`: the stored (normalised) fp32 rows. */
int rdx_index_get(rdx_index* h, const int64_t* row_ids, int64_t n, float* out, int space);
/* `collection.delete(ids=...)` (reference ingest_enterprise.py:272,304): keep exactly the rows
 * listed in `keep` (strictly ascending, host pointer), renumbering them 0..n_keep-1. */
int rdx_index_compact(rdx_index* h, const int64_t* keep, int64_t n_keep);

/* Shards whose rows are not one contiguous range of the collection (several GPUs behind ONE `collection` object, rows
 * dealt to the devices as they arrive): returned row id of local row r = ids[r]. ids must be strictly increasing over the
 * shard (ties inside a shard are ordered by local row, which then is the order of the returned ids too). Sets the ids
 * of rows [first_row, first_row + n), which must exist; rows never given an id return local + "row_base". An index
 * without a map (the default) returns local + "row_base". rdx_index_compact drops the map (rows are renumbered). */
int rdx_index_set_row_ids(rdx_index* h, int64_t first_row, const int64_t* ids, int64_t n, int space);

/* Options (tests and benchmarks): "force_exact" 0/1, "force_fast" 0/1 (MFMA scan even for small
 * problems), "sample_div" >=1, "cand_cap" 0 (auto) or slots per (query, stream) candidate segment,
 * "profile" 0..3: 1 = HIP events around every kernel of the next searches, 2 = two events around the dominant kernel(s) only
 * (main scan, or K5a + K5b on the exact path), 3 = no events: those kernels stamp their own first-workgroup start and
 * last-workgroup end (an event record costs the stream ~6 us, 10 % of a small search) — rdx_search_stats.ms_*; "row_base" >= 0:
 * added to every returned row id, so a shard holding rows [base, base+count) answers with GLOBAL ids;
 * "sib_sync" 0/1 (default 0) and "sib_lag" 3..100 (k-steps): soft lock-step of the workgroups that stream
 * the same corpus tiles for different query tiles (less fabric traffic for ~2-3 % of the throughput while the
 * launch is MFMA-bound; speed and traffic only, never results); "xcd_balance" 0/1 (default 1): the main scan's
 * tiles are split between the 8 XCDs by their measured speed in the previous searches instead of evenly (the XCDs of
 * one chip differ by up to 10 %; speed only); "compact_master" 0/1 (default 0; only while the index is empty): the exact copy of the rows keeps the raw bf16 rows as
 * delivered by rdx_index_add_bf16 plus one fp64 divisor per row (2 B/element + 8 B/row) instead of the normalised fp32 rows
 * (4 B/element); every normalised element is recomputed as (float)((double)x / den) where the exact re-score, the exact scan
 * and rdx_index_get need it — the operands and the operation of the ingest normalisation, hence the same bits. With the fp16
 * scan copy that is 4 instead of 6 B/element for a bf16 corpus (BASELINE config 5). Such an index takes rows through
 * rdx_index_add_bf16 only (rdx_index_add / _add_stored / _update return RDX_ERR_STATE);
 * "fuse_epilogue" 0/1 (default 1): B > 128 main scan variant whose per-tile emit check rides inside the
 * first k-step of the next tile instead of interrupting the MFMA stream (speed only: +1 % at B = 1024; used when the
 * number of 64-element k-steps per row is even, the stand-alone check otherwise and with 0);
 * "wave_layout" 0/1 (default 0; developer experiment): 1 = the fused B > 128 main scan of launches with several query tiles runs
 * one wave per SIMD, each owning 64 rows x 256 queries (csrc/scan_w4.hpp) instead of two owning 32 x 256 (speed only);
 * "spec_tau" 0/1 (default 1): the scan threshold is taken from a rank below k of the sampled scores — an estimate of the corpus'
 * k-th score instead of a proven lower bound — and verified per query afterwards (c_k - 2E >= threshold); a query that fails
 * takes the fallback passes with the proven threshold (speed only: 2-6x fewer candidates; never results);
 * "split_boot" 0/1 (default 1): searches of <= 64 queries whose threshold sample is small take it with the split-K bootstrap
 * kernel (one 32-row block per workgroup, the k-steps dealt to its waves) instead of whole tiles on a few CUs;
 * "small_scan" 0/1 (default 1): the same searches on corpora of at most 32 such blocks per CU (262 K rows on 256 CUs) take their
 * main scan with the split-K kernel too (balanced in 32-row units instead of whole 256-row tiles);
 * "half_boot" 0/1 (default 1): searches of 129..256 queries take their threshold sample as two 128-query tiles per sampled corpus
 * tile (every CU busy, less data per k-step) instead of one 256-query tile on half the CUs;
 * "spread_boot" 0/1 (default 1): searches of more than 64 queries take their threshold sample as every div-th 32-row block of the corpus
 * instead of every div-th 256-row tile (the same number of rows, eight times finer: runs of similar rows stored together — a
 * document's chunks — are met by the sample instead of falling between two sampled tiles; speed only);
 * "fuse_finish" 0/1 (default 1): the end-of-search work (counters and small results to pinned host memory) runs in the last
 * block of the search's last kernel instead of a launch of its own (both: speed only);
 * "retry" 0/1 (default 1): queries whose candidate
 * segments overflow get a second MFMA pass as a small batch (denser threshold sample) before the exact full scan. */
int rdx_index_set_option(rdx_index* h, const char* name, int64_t value);

/* The main scan's tile shares of the 8 XCDs (1.0 = an eighth; option "xcd_balance"): learned from the workgroups' own time stamps
 * over the first searches of a process — the first launches of a cold index run with even shares, ~15 % slower on a 10 M-row
 * scan. out8 (may be NULL) receives the current shares; in8 (may be NULL) replaces them (each clamped to [0.6, 1.5], renormalised
 * to sum 8): a host layer that persists an index (rag_dpo_amd/collection.py) stores them with it and hands them back at load.
 * Speed only, never results; they re-adapt if the values no longer fit the device. */
int rdx_index_xcd_shares(rdx_index* h, double* out8, const double* in8);

/* `SentenceTransformer.encode(..., normalize_embeddings=True)`'s last step
 * (reference src/utils/embedding_provider.py:139-145): out[i] = in[i] / max(||in[i]||_2, 1e-12). */
int rdx_l2_normalize(int device, const float* in, int64_t n, int dim, float* out, int space,
                     void* stream);

/* Two fused kernels for the memory-bound parts of the query encoder's forward — what `SentenceTransformer.encode` runs on the
 * GPU in front of that normalisation (reference src/utils/embedding_provider.py:118-147; the GEMMs stay the BLAS library's, driven
 * from rag_dpo_amd/embedding_provider.py). fp16 device pointers, 16-byte aligned; enqueued on `stream`, nothing is synchronised.
 *
 * rdx_enc_attention_f16: self-attention over PACKED tokens (no padding). qkv [n_tokens][3*heads*head_dim]: per token its query,
 *   key and value rows (heads x head_dim each, head-major); token t belongs to the text whose tokens are
 *   tok_first[t] .. tok_first[t] + tok_len[t] - 1 (tok_len >= 1) and attends to exactly those. ctx [n_tokens][heads*head_dim] =
 *   softmax(q k^T * scale) v per head, soft-max and accumulation in fp32. head_dim must be 64. Meant for short texts
 *   (questions): the work per token grows with its text's length. max_text_tokens: the longest text's token count if the caller
 *   knows it (sizes the LDS window a workgroup stages keys in; a wrong or unknown value — pass 0 — costs speed only: rows whose
 *   text does not fit the window read global memory).
 * rdx_enc_add_layernorm_f16: out[r] = LayerNorm(a[r] + b[r]) * gamma + beta over rows of `hidden` halves (512, 1024, 1536 or
 *   2048; biased variance, fp32 statistics, the sum rounded to fp16 first — what an fp16 add followed by LayerNorm computes). */
int rdx_enc_attention_f16(int device, const void* qkv, const int32_t* tok_first, const int32_t* tok_len,
                          int64_t n_tokens, int heads, int head_dim, float scale, int max_text_tokens, void* ctx,
                          void* stream);
/* rdx_enc_attention_mfma_f16: the same attention for texts of ANY length (the corpus side: chunk texts of up to ~1 K tokens, reference
 *   src/processing/create_chromadb_index.py:300-387) on the matrix cores, flash-style. query_blocks [n_blocks][4] int32 =
 *   {first token of the text, its length, first query of this block within the text, 0}: one workgroup per block of <= 64 queries and
 *   head; the caller lists ceil(length / 64) blocks per text. Same arithmetic contract as rdx_enc_attention_f16 (soft-max and
 *   accumulation in fp32; probabilities rounded to fp16 before the PV product, as a flash kernel does). */
int rdx_enc_attention_mfma_f16(int device, const void* qkv, const int32_t* query_blocks, int n_blocks, int heads,
                               int head_dim, float scale, void* ctx, void* stream);
int rdx_enc_add_layernorm_f16(int device, const void* a, const void* b, const void* gamma, const void* beta,
                              float eps, int64_t rows, int hidden, void* out, void* stream);
/* rdx_enc_gelu_f16: x[i] = gelu(x[i]) IN PLACE over n fp16 values (n a multiple of 8, x 16-byte aligned): the erf form the
 *   checkpoint's FFN uses (HF XLMRobertaIntermediate, hidden_act "gelu"), fp32 arithmetic, erf by Abramowitz & Stegun 7.1.26
 *   (|error| <= 1.5e-7): the framework's fp16 result or its neighbour; within 1e-6 absolutely where the result underflows. */
int rdx_enc_gelu_f16(int device, void* x, int64_t n, void* stream);
/* rdx_enc_linear_small_f16: out[n_tokens][n_out] = act(x[n_tokens][n_in] w[n_out][n_in]^T + bias[n_out]) for at most 256 tokens (one
 *   question, a question's sub-queries): the weight matrix is read once, fp32 accumulation; act 0 = none, 1 = erf GELU (the
 *   checkpoint's). n_out a multiple of 16, n_in of 512. Larger token counts belong to the BLAS library. */
int rdx_enc_linear_small_f16(int device, const void* x, const void* w, const void* bias, int n_tokens, int n_out,
                             int n_in, int act, void* out, void* stream);

/* The forward of ONE question — `EmbeddingProvider.embed_query` / `embed([q])` in front of every `collection.query`
 * (reference src/utils/embedding_provider.py:118-157, called at src/rag/retriever.py:150-154, 212, 377) — as five launches per
 * transformer layer for at most 32 packed tokens (a question, or a question's short sub-queries). All pointers fp16 device
 * memory, 16-byte aligned, enqueued on `stream`, nothing synchronised. rag_dpo_amd/embedding_provider.py chains them and replays
 * the chain as one HIP graph.
 *
 * rdx_enc_embed_f16: out[t] = (word[tok[t]] + pos[pos_id[t]]) + type0 — the embedding sum in the module's order of fp16 adds; its
 *   LayerNorm is the first rdx_enc_stage_f16's prologue.
 * rdx_enc_stage_f16: out[n_tokens][n_out] = epi(in[n_tokens][n_in] w[n_out][n_in]^T + bias), fp32 accumulation, the weight matrix
 *   read once by n_out / features_per_workgroup workgroups (16, or 8 / 4 so that a 1024-feature projection still covers 128 / 256
 *   CUs; 0 = 16). n_in 512, 1024, 2048 or 4096.
 *     ln_gamma != NULL: `x` holds pre-LayerNorm sums s and in = LayerNorm(s) * ln_gamma + ln_beta (fp32 statistics, biased
 *       variance), recomputed by every workgroup; if y_out != NULL that LayerNorm output [n_tokens][n_in] is stored as well (the
 *       residual of the block's second half). n_in 512 or 1024, 16 features per workgroup, epilogue 0 or 1.
 *     ln_gamma == NULL: in = x, or rows x_rows[t] of x (and of `res`) when x_rows != NULL (the last layer: CLS rows only).
 *     epilogue 0: + bias; 1: erf GELU(+ bias); 2: res + (fp16)(+ bias) — the block's residual sum, rounded as the module's
 *       fp16 add rounds it; `out` then holds the next LayerNorm's input.
 * rdx_enc_attention_small_f16: rdx_enc_attention_f16 for at most 32 tokens on MFMA, one workgroup per head: token t attends to
 *   the tokens u with tok_first[u] == tok_first[t].
 * rdx_enc_layernorm_rows_f16: out[r] (fp32) = (fp16) LayerNorm(s[r]) — the last LayerNorm, on the CLS rows. */
int rdx_enc_embed_f16(int device, const int64_t* tok, const int64_t* pos_id, const void* word, const void* pos,
                      const void* type0, int n_tokens, int hidden, void* out, void* stream);
int rdx_enc_stage_f16(int device, const void* x, const int64_t* x_rows, const void* ln_gamma, const void* ln_beta,
                      float ln_eps, void* y_out, const void* w, const void* bias, const void* res, int n_tokens,
                      int n_out, int n_in, int epilogue, int features_per_workgroup, void* out, void* stream);
int rdx_enc_attention_small_f16(int device, const void* qkv, const int32_t* tok_first, int n_tokens, int heads,
                                int head_dim, float scale, void* ctx, void* stream);
int rdx_enc_layernorm_rows_f16(int device, const void* s, const void* gamma, const void* beta, float eps, int rows,
                               int hidden, float* out, void* stream);

/* `collection.query(query_embeddings=, n_results=k, where=)` (reference
 * src/rag/retriever.py:215-220,380-385; create_chromadb_index.py:405-408,435-439).
 *   queries     [nq][dim] raw fp32 (normalised on the device like corpus rows)
 *   allow_bits  NULL, or ceil(count/32) words: bit (r&31) of word r>>5 set = row r may be
 *               returned (the `where` pre-filter and tombstones, evaluated by the host layer)
 *   out_score   [nq][k] fp32 cosine, out_row [nq][k] int64 row ids, out_count [nq] number of valid
 *               entries (= min(k, allowed rows)); unused tail entries are (-inf, -1). */
int rdx_search(rdx_index* h, const float* queries, int64_t nq, int k, const uint32_t* allow_bits,
               float* out_score, int64_t* out_row, int32_t* out_count, int space, void* stream);

/* Resident `where` bitmaps. The reference's filters are a handful of fixed shapes (src/rag/pipeline.py:35-71,
 * pages/1_Chat.py:245-247) sent with every question: the host layer evaluates one ONCE, keeps the bitmap in HBM as an
 * rdx_mask and passes the handle with every search instead of re-uploading count/8 bytes. allow_bits as in rdx_search
 * (ceil(count/32) words, host or device). A mask belongs to the row count it was made for: any add / compact makes
 * rdx_search_masked refuse it (RDX_ERR_STATE) — rebuild it (update of a row's vector does not change the count; the host
 * layer drops its masks on every write anyway). mask == NULL = no filter. */
typedef struct rdx_mask rdx_mask;
int rdx_mask_create(rdx_index* h, const uint32_t* allow_bits, int space, rdx_mask** out);
int rdx_mask_destroy(rdx_mask* m);
int rdx_search_masked(rdx_index* h, const float* queries, int64_t nq, int k, const rdx_mask* mask, float* out_score,
                      int64_t* out_row, int32_t* out_count, int space, void* stream);

/* The search in two halves, for callers that enqueue consumers of the results on the same stream (the multi-GPU path: the
 * RCCL all-gather of the partial top-k and the merge, reference has no counterpart). rdx_search_async enqueues the whole
 * search (device pointers only, nq <= 4096; mask may be NULL) and returns without waiting; rdx_search_wait blocks until that
 * search — not the stream: work enqueued behind it, e.g. the next batch's query encode, is not waited for — has completed and
 * runs its host half: when some queries' candidate segments overflowed (rare: clustered corpora) it
 * re-runs them through the fallback passes, synchronises the stream and reports *redone = 1 — results written by the first
 * pass were incomplete for those queries, so whatever consumed them on the stream must be re-enqueued. Any other call on
 * the index completes a pending search first. rdx_search == rdx_search_async + rdx_search_wait for device callers.
 * Lifetimes: `queries` is consumed by the kernels rdx_search_async enqueues (the fallback passes work from the index's own
 * normalised copy), so the caller may overwrite or free it with any STREAM-ORDERED work enqueued afterwards; the three output
 * buffers and out_flags must stay valid until rdx_search_wait has returned (the host half may rewrite them).
 * out_flags: NULL, or int32[RDX_PACKED_FLAGS] on the device. The search's last kernel stores out_flags[0] = 1 when the
 * results are incomplete and rdx_search_wait WILL redo them (exactly the cases in which it reports *redone = 1), else 0;
 * [1..3] = 0. rdx_search_wait stores 0 again once the fallback passes are enqueued: a consumer ordered after it sees
 * "complete". With the partial laid out as below the word travels with the all-gather, and every rank learns from the
 * merge (rdx_signal) whether ANY rank's partial was incomplete — no second collective, no device-to-host copy per step. */
#define RDX_PACKED_FLAGS 4
int rdx_search_async(rdx_index* h, const float* queries, int64_t nq, int k, const rdx_mask* mask, float* out_score,
                     int64_t* out_row, int32_t* out_count, int32_t* out_flags, void* stream);
int rdx_search_wait(rdx_index* h, int* redone);

/* Multi-GPU exchange step: merge n_parts per-shard partial results (after the RCCL all-gather,
 * SURVEY.md §8e) into the global top-k with the same ordering rule. Layouts:
 * part_score/part_row [n_parts][nq][k], part_count [n_parts][nq]; row ids must already be global (no row in two parts).
 * n_parts <= 64, k <= 4096; when n_parts * k exceeds 4096 the parts are folded pairwise (same result, more launches). */
int rdx_merge_topk(int device, const float* part_score, const int64_t* part_row,
                   const int32_t* part_count, int n_parts, int64_t nq, int k, float* out_score,
                   int64_t* out_row, int32_t* out_count, int space, void* stream);

/* A word in pinned host memory that a kernel publishes and the host waits on (spin, then stream synchronise): how the
 * merge tells the host whether any rank's partial carried the "incomplete" flag. One signal serves one stream of merges. */
typedef struct rdx_signal rdx_signal;
int rdx_signal_create(int device, rdx_signal** out);
int rdx_signal_destroy(rdx_signal* s);
/* blocks until the LAST rdx_merge_topk_packed given `s` has published; *value = OR over the parts of flags[0]. `stream` = the
 * stream that merge was enqueued on: never synchronised (work enqueued behind the merge is not waited for) — the word is spun
 * on for ~0.4 ms, then polled between short sleeps, and the stream is only queried now and then to report a lost merge. */
int rdx_signal_wait(rdx_signal* s, void* stream, int32_t* value);

/* Same merge, reading the partials straight out of the all-gather receive buffer (device memory):
 * part p starts part_stride bytes (multiple of 16) after part p-1 and is one rank's packed contribution
 * rows int64[nq][k] | scores f32[nq][k] | counts int32[nq] | flags int32[RDX_PACKED_FLAGS]
 * (= nq*k*12 + nq*4 + 16 bytes). Outputs are device pointers; k >= 1; n_parts * k <= 4096 (unlike rdx_merge_topk the packed form
 * does not fold: rag_dpo_amd/sharded.py checks world * k before it enqueues anything, so no rank fails behind a collective). sig: NULL, or the signal through which the first block
 * publishes the OR of the parts' flags[0] as soon as it has read them (the decision "some partial was incomplete: exchange
 * again" does not wait for the merge itself). */
int rdx_merge_topk_packed(int device, const void* packed, int64_t part_stride, int n_parts, int64_t nq,
                          int k, float* out_score, int64_t* out_row, int32_t* out_count, rdx_signal* sig, void* stream);

/* Diagnostics of the last rdx_search on this index (valid after the stream has synchronised). */
typedef struct rdx_search_stats {
    int64_t nq, k, rows;
    int64_t sample_rows;      /* rows scanned by the threshold bootstrap pass */
    int64_t emitted;          /* candidates emitted by the main scan (sum over queries) */
    int64_t rescored;         /* candidates re-scored exactly (sum over queries) */
    int64_t exact_queries;    /* queries answered by the exact full scan (fallback / small path) */
    int32_t path;             /* 0 = MFMA scan + exact re-score, 1 = exact scan only */
    int32_t profiled;         /* 1 if the ms_* fields below were measured */
    float ms_normalize, ms_scan_sample, ms_tau, ms_scan_main, ms_refine, ms_exact, ms_total;
    int64_t scan_main_launch_rows, scan_main_launch_queries; /* units of the dominant kernel */
    int64_t retried_queries;  /* queries whose candidate segments overflowed and that got a second MFMA pass */
    float xcd_finish_spread_ms; /* main scan: last XCD's finish minus first XCD's finish (0 when not measured) */
    float xcd_share_min, xcd_share_max; /* smallest / largest XCD share of the tiles (1.0 = an eighth) used by that scan */
    float tau_rank;           /* rank of the sampled score the scan threshold was taken from: k = provable, < k = speculative (verified per query) */
} rdx_search_stats;
int rdx_search_last_stats(rdx_index* h, rdx_search_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* RDX_H */
