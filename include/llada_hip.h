/*
 * llada_hip.h -- C ABI of libllada_hip.so: the MI355X (gfx950) implementation of the LLaDA
 * masked-diffusion sampling loop, its masked log-likelihood estimator and the non-causal
 * LLaDA-8B forward they call.
 *
 * The reference (taiqihe/LLaDA) has no FFI: its boundary is two Python callables and a duck-typed
 * model (generate.py:43-45, get_log_likelihood.py:46-47, model(x, attention_mask).logits at
 * generate.py:88,92 and get_log_likelihood.py:38).  Each entry point below names the reference
 * lines whose ATen op sequence it replaces.  The Python host in llada_amd/ keeps the reference
 * signatures and calls these functions through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host"; the library never
 *    allocates, frees or keeps caller memory; scratch comes in through `workspace` arguments.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*) and never
 *    synchronises the device.
 *  - bf16 tensors are raw 16-bit patterns (uint16_t), row-major.
 *  - return value: LLADA_OK or a negative LLADA_ERR_*; llada_last_error() gives the text of
 *    the calling thread's most recent failure.
 */
#ifndef LLADA_HIP_H
#define LLADA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLADA_ABI_VERSION 4

#define LLADA_OK 0
#define LLADA_ERR_INVALID (-1)      /* bad argument (the Python host raises AssertionError/ValueError) */
#define LLADA_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels are built for */
#define LLADA_ERR_HIP (-3)          /* a HIP runtime call failed */
#define LLADA_ERR_WORKSPACE (-4)    /* workspace too small */

/* remasking rules of generate.py:103-123 */
#define LLADA_REMASK_LOW_CONFIDENCE 0
#define LLADA_REMASK_HIGH_CONFIDENCE 1
#define LLADA_REMASK_LOW_ENTROPY 2
#define LLADA_REMASK_HIGH_ENTROPY 3
#define LLADA_REMASK_RANDOM 4

/* flags of llada_remask_scores */
#define LLADA_FLAG_LOGITS_EOS_INF 1u       /* generate.py:94-95 */
#define LLADA_FLAG_CONF_EOS_EOT_INF 2u     /* generate.py:100-101 */
#define LLADA_FLAG_CONF_FLOAT32 4u         /* confidence = float32 softmax probability, not rounded to bf16:
                                              eval_llada.py:187 (suffix_greedy_prediction) */

/* GEMM epilogues */
#define LLADA_EPI_PLAIN 0      /* C = bf16(acc) */
#define LLADA_EPI_RESIDUAL 1   /* C = bf16(R + bf16(acc)) */
#define LLADA_EPI_ROPE_QKV 2   /* QKV projection with rotary embedding applied to Q and K (packed weight layout) */
#define LLADA_EPI_SWIGLU 3     /* gate/up projection, C[m][f] = bf16(bf16(silu(gate)) * up) (packed weight layout) */
/* ORed into the epilogue argument: W is stored tile-major -- every 256-row x 64-column block of the [N, K]
 * matrix contiguous (256 rows of 128 bytes), the K/64 blocks of one 256-row band consecutive, bands in row
 * order: element (n, k) at ((n/256 * K/64 + k/64) * 256 + n%256) * 64 + k%64.  One K step of one output tile
 * then reads one contiguous 32 KiB run instead of 256 segments of 128 bytes at a stride of 2K bytes.
 * Needs N % 256 == 0. */
#define LLADA_GEMM_W_TILED 0x100

int llada_abi_version(void);
const char* llada_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Forward pass of the mask predictor: replaces `model(x, attention_mask=...).logits`
 * (generate.py:88,92; get_log_likelihood.py:38).  The model code itself is not part of the
 * reference tree (HF remote code); the architecture is restated from GUIDELINES.md:6-16.
 * ------------------------------------------------------------------------------------------- */

typedef struct llada_config {
    int32_t d_model;      /* 4096 */
    int32_t n_layers;     /* 32 */
    int32_t n_heads;      /* 32; head_dim = d_model / n_heads must be 128 */
    int32_t mlp_hidden;   /* 12288 */
    int32_t vocab_size;   /* 126464 */
    float rms_eps;        /* 1e-5 */
} llada_config;

/* Packed weights (see llada_amd/weights.py for the packing):
 *   wqkv[l]  [3*d, d]   rows = Q heads | K heads | V heads; inside every Q and K head the 128
 *                       rows are stored in the order 0-15,64-79,16-31,80-95,32-47,96-111,
 *                       48-63,112-127 so that a rotary pair (j, j+64) meets in one lane of the
 *                       GEMM epilogue (attention is invariant to a common permutation of the
 *                       Q and K head dimensions).
 *   wgu[l]   [2*F, d]   gate and up rows interleaved in blocks of 16: gate 0-15, up 0-15,
 *                       gate 16-31, up 16-31, ...
 *   wo[l] [d, d], wdown[l] [d, F], lm_head [V, d], wte [V, d]: as in the checkpoint.
 * The per-layer arrays are HOST arrays of n_layers device pointers.
 * packed is a bit set.  Bit 0 clear selects the checkpoint layout instead (wqkv = q|k|v rows and wgu =
 * gate|up rows in natural order); the forward then runs the rotary embedding and SwiGLU as separate kernels.
 * Bit 1 (value 2): wqkv, wo, wgu, wdown and lm_head are stored tile-major (LLADA_GEMM_W_TILED; the row order
 * above is applied first); wte stays row-major. */
typedef struct llada_weights {
    int32_t packed;
    const void* wte;
    const void* ln_f;
    const void* lm_head;
    const void* const* attn_norm;
    const void* const* wqkv;
    const void* const* wo;
    const void* const* ff_norm;
    const void* const* wgu;
    const void* const* wdown;
} llada_weights;

/* Bytes of scratch llada_forward needs for `n_seq` sequences of `seq_len` tokens and
 * `n_logit_rows` LM-head rows. */
size_t llada_forward_workspace_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len, int32_t n_logit_rows);

/* ids [n_seq, seq_len] int64; key_mask [n_seq, seq_len] uint8 (1 = attend) or NULL;
 * rope_cos / rope_sin [seq_len, 64] float32; logit_rows [n_logit_rows] int32 flat token
 * indices (seq * seq_len + pos) whose logits are wanted, or NULL for all tokens in order;
 * logits_out [n_logit_rows, vocab] bf16.
 * On return (stream-ordered) the first n_seq * seq_len * d_model bf16 of `workspace` hold the residual stream
 * after the last layer, before the final norm (a model's last `hidden_states`), provided every token's logits
 * were asked for (logit_rows == NULL); with a row subset the last layer updates only those rows elsewhere. */
int llada_forward(const llada_config* cfg, const llada_weights* w, const int64_t* ids, const uint8_t* key_mask,
                  const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                  const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out, void* workspace,
                  size_t workspace_bytes, void* stream);

/* llada_forward with the rotary position of every token given by the caller: positions [n_seq * seq_len] int32,
 * the row of rope_cos / rope_sin each token is rotated with (values outside [0, seq_len) are clamped), or NULL for
 * the token's index inside its sequence.  The reference model always rotates by the column index (`arange(L)`
 * whatever the padding), and rotary attention depends on position differences only, so a left-padded row gives the
 * same logits up to rounding with either numbering; numbering every row from its first real token (together with a
 * key_mask over the padding) makes the rotated queries and keys of a row in a left-padded batch the very values the
 * row has when it is run alone -- what a caller needs to batch requests that /root/reference/eval_llada.py:260-265
 * runs one at a time and still reproduce their results. */
int llada_forward_positions(const llada_config* cfg, const llada_weights* w, const int64_t* ids, const uint8_t* key_mask,
                            const int32_t* positions, const float* rope_cos, const float* rope_sin, int32_t n_seq,
                            int32_t seq_len, const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ----- the forward's building blocks, exported for per-kernel parity tests and benches ----- */

/* out[m, :] = wte[ids[m], :]  (embedding lookup of the forward) */
int llada_embed_bf16(const void* wte, const int64_t* ids, void* out, int64_t n_tokens, int32_t d, int64_t vocab,
                     void* stream);

/* out[r, :] = in[rows[r], :], in [n_in, d] bf16.  The forward uses it in the last layer, where only the
 * rows whose logits are wanted (generate.py:125-128 discards the rest; get_log_likelihood.py:40
 * indexes the masked answer positions) go through the output projection and the MLP. */
int llada_gather_rows_bf16(const void* in, const int32_t* rows, void* out, int64_t n_rows, int32_t d, int64_t n_in,
                           void* stream);

/* RMSNorm: out[r, :] = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w), float32 inside; if row_index
 * is not NULL, output row r reads input row row_index[r] (gather fused into the final norm). */
int llada_rmsnorm_bf16(const void* x, const void* w, void* out, int64_t n_rows, int32_t d, float eps,
                       const int32_t* row_index, void* stream);

/* C[M, N] = A[M, K] @ W[N, K]^T on bf16 MFMA with float32 accumulation.  lda/ldc/ldr are row
 * strides in elements.  K % 64 == 0 and N % 8 == 0 are required.  aux: residual R [M, N]
 * (EPI_RESIDUAL, row stride ldr), or NULL.  For EPI_ROPE_QKV pass rope_cos/rope_sin
 * [seq_len, 64] float32, seq_len and d_model (= N / 3); for EPI_SWIGLU C is [M, N/2].
 * workspace (may be NULL): float32 scratch for split-K partial sums; when M is so small that the
 * output tiles would leave most CUs idle, K is cut into ranges summed by a second pass, in a
 * fixed order.  llada_gemm_workspace_bytes gives the size that removes every limit. */
int llada_gemm_bf16(const void* A, const void* W, void* C, int64_t M, int32_t N, int32_t K, int32_t lda,
                    int32_t ldc, int32_t epilogue, const void* aux, int32_t ldr, const float* rope_cos,
                    const float* rope_sin, int32_t seq_len, void* workspace, size_t workspace_bytes,
                    void* stream);
size_t llada_gemm_workspace_bytes(int64_t M, int32_t N, int32_t K);

/* X[M, N] += A[M, K] @ W[N, K]^T (bf16 residual stream, in place, row stride N) and
 * H[M, N] = RMSNorm(X) * norm_w: a residual projection of the mask predictor (attn_out or ff_out, called at
 * generate.py:88,92 / get_log_likelihood.py:38) together with the norm that reads its result.  Bit-identical
 * to llada_gemm_bf16(EPI_RESIDUAL) followed by llada_rmsnorm_bf16; when the GEMM runs split-K (small M) the
 * second pass writes both outputs, which saves a kernel and a round trip of the residual stream.
 * w_tiled != 0: W is tile-major (LLADA_GEMM_W_TILED). */
int llada_gemm_residual_rmsnorm_bf16(const void* A, const void* W, void* X, void* H, const void* norm_w, float eps,
                                     int64_t M, int32_t N, int32_t K, int32_t lda, int32_t w_tiled, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* In-place rotary embedding of the Q and K thirds of qkv [n_seq*seq_len, 3*d] (natural head
 * layout, pairs (j, j+64)); float32 arithmetic, bf16 result. */
int llada_rope_bf16(void* qkv, const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                    int32_t n_heads, void* stream);

/* out[m, f] = bf16(bf16(silu(gu[m, f])) * gu[m, F + f]) for gu [n_rows, 2F] (natural layout). */
int llada_swiglu_bf16(const void* gu, void* out, int64_t n_rows, int32_t F, void* stream);

/* Bidirectional (no causal mask) multi-head attention, head_dim 128.  qkv [n_seq*seq_len, 3*d]
 * (Q | K | V thirds, head h at columns h*128), key_mask [n_seq, seq_len] uint8 or NULL,
 * out [n_seq*seq_len, d].  softmax(q k^T / sqrt(128)) v with float32 online softmax. */
int llada_attn_bidir_bf16(const void* qkv, const uint8_t* key_mask, void* out, int32_t n_seq, int32_t seq_len,
                          int32_t n_heads, void* stream);

/* The same attention for a subset of the positions: q [n_seq*n_q, q_stride] (n_q query rows per sequence,
 * head h at columns h*128) over all seq_len keys/values of a [Q | K | V] image kv [n_seq*seq_len, 3*d];
 * out [n_seq*n_q, d].  block_kv (may be NULL): compact [Q | K | V] rows [n_seq*n_q, 3*d] whose K and V thirds
 * are used in place of the image's rows block_start .. block_start+n_q-1 of every sequence (the image is only
 * read).  Used by the block-cached sampler (llada_forward_block). */
int llada_attn_bidir_queries_bf16(const void* q, int64_t q_stride, int32_t n_q, const void* kv, const uint8_t* key_mask,
                                  void* out, int32_t n_seq, int32_t seq_len, int32_t n_heads, const void* block_kv,
                                  int32_t block_start, void* workspace, size_t workspace_bytes, void* stream);
/* Float32 scratch for the call above (workspace may be NULL): with it, when n_seq * n_heads * ceil(n_q / 128)
 * workgroups would leave most CUs idle, the keys are cut into runs of 64-key tiles, one workgroup each, whose
 * unnormalised partial results a second kernel merges (same softmax, different summation order). */
size_t llada_attn_workspace_bytes(int32_t n_seq, int32_t n_q, int32_t seq_len, int32_t n_heads);

/* ---------------------------------------------------------------------------------------------
 * Block-cached forward.  No counterpart in the reference, whose sampler recomputes every position at
 * every step (generate.py:80-92; README.md:110-116 names the missing cache as the reason it is slow
 * and points to Fast-dLLM / dLLM-Cache, README.md:127-130).  An opt-in APPROXIMATION of the sampler:
 * llada_forward_fill_cache is llada_forward that also leaves every layer's [Q | K | V] image
 * ([n_layers][n_seq*seq_len][3*d] bf16, llada_kv_cache_bytes) in kv_cache; llada_forward_block then
 * recomputes only the block_len positions from block_start of every sequence (block_ids
 * [n_seq, block_len]): their keys/values replace the cached ones, their queries attend over the whole
 * cached sequence, and logits_out [n_seq*block_len, vocab] holds their logits -- or, with logit_rows
 * ([n_logit_rows] int32 indices into the n_seq*block_len recomputed rows, seq * block_len + offset), only the
 * logits of those rows, logits_out [n_logit_rows, vocab].  Keys and values of the
 * other positions stay as they were when the cache was filled.  The range may reach to the end of the sequence
 * (block_start + block_len == seq_len: everything right of a cached prefix is recomputed).  update_cache == 0 leaves the cache untouched
 * (the block's fresh keys/values are used for this call only: enough when the next call for another block is
 * preceded by a new llada_forward_fill_cache, as in generate()); the logits are the same either way.
 * ------------------------------------------------------------------------------------------- */
size_t llada_kv_cache_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len);
int llada_forward_fill_cache(const llada_config* cfg, const llada_weights* w, const int64_t* ids,
                             const uint8_t* key_mask, const float* rope_cos, const float* rope_sin, int32_t n_seq,
                             int32_t seq_len, const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out,
                             void* workspace, size_t workspace_bytes, void* kv_cache, void* stream);
size_t llada_forward_block_workspace_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len, int32_t block_len);
int llada_forward_block(const llada_config* cfg, const llada_weights* w, const int64_t* block_ids,
                        const uint8_t* key_mask, const float* rope_cos, const float* rope_sin, int32_t n_seq,
                        int32_t seq_len, int32_t block_start, int32_t block_len, void* kv_cache, int32_t update_cache,
                        const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sampler epilogue: replaces generate.py:81-134 after the forward.
 * ------------------------------------------------------------------------------------------- */

/* generate.py:22-40 + :75-79: per row, count the mask tokens in columns [block_start, block_end)
 * of x [B, L] and write the schedule sched[b, 0..steps) (int32): n/steps, +1 on the first
 * n%steps steps. */
int llada_num_transfer_tokens(const int64_t* x, int32_t B, int32_t L, int32_t block_start, int32_t block_end,
                              int64_t mask_id, int32_t steps, int32_t* sched, void* stream);

/* add_gumbel_noise (generate.py:8-19) as a map of its own, for callers that import the helper: out[r, v] =
 * exp(logits[r, v]) / (-log u[r, v]) ** temperature in float64, logits / out [n_rows, vocab] float64 (the
 * reference converts to float64 first, :16), row r = batch row r / L, position r % L.  noise [n_rows, vocab]
 * float64 uniforms replace torch.rand_like (:17), or NULL = the library's Philox stream with the counters
 * llada_remask_scores uses for (seed, step, row_offset), so argmax(out) is that call's candidate token.
 * generate() never calls this: its epilogue ranks the same scores in registers. */
int llada_add_gumbel_noise(const double* logits, const double* noise, double temperature, double* out, int64_t n_rows,
                           int32_t vocab, int32_t L, uint64_t seed, uint32_t step, uint32_t row_offset, void* stream);

typedef struct llada_remask_args {
    const void* logits;        /* [n_rows, vocab] bf16 (the conditional half under CFG) */
    const void* uncond_logits; /* [n_rows, vocab] bf16 or NULL; generate.py:89-90 combine is fused */
    double cfg_scale;          /* (cfg_scale + 1) is formed in double, then used as a float32 scalar */
    const int32_t* row_pos;    /* [n_rows] flat index b*L + column of each logits row, or NULL = identity */
    int32_t n_rows;
    int32_t vocab;
    const int64_t* x;          /* [B, L] current tokens */
    int32_t B, L;
    int32_t block_end;         /* P + (num_block+1)*block_length, generate.py:125 */
    int64_t mask_id;
    int32_t remasking;         /* LLADA_REMASK_* */
    uint32_t flags;            /* LLADA_FLAG_* */
    int32_t eos_id, eot_id;    /* 126081, 126348 */
    float temperature;
    const double* noise;       /* [n_rows, vocab] float64 uniforms replacing torch.rand_like
                                  (generate.py:17), or NULL = the library's Philox stream */
    const float* rand_scores;  /* [B, L] float32 replacing torch.rand (generate.py:121), or NULL */
    uint64_t seed;             /* Philox key */
    uint32_t step;             /* global denoise step (Philox counter) */
    uint32_t row_offset;       /* global index of batch row 0 (data-parallel shards) */
    int64_t* x0;               /* out [B, L]: candidate tokens, generate.py:127 */
    float* confidence;         /* out [B, L]: generate.py:128 (bf16 values except for 'random') */
    void* workspace;           /* llada_rows_workspace_bytes(n_rows) bytes, 16-byte aligned, or NULL */
    size_t workspace_bytes;
} llada_remask_args;

/* Fused generate.py:81,89-90,94-128: one pass over the logits per active row gives the argmax
 * (float64 Gumbel-max when temperature > 0), the softmax statistics and the confidence; rows
 * that are not mask tokens or lie beyond block_end get x0 = x, confidence = -inf, as do all
 * positions no logits row covers.
 * With a workspace, temperature 0 and a rule other than the two entropy rules, every row is cut into
 * up to 8 pieces streamed by one small workgroup each and merged by a second kernel (a few rows then
 * still fill the chip; any vocabulary size that is a multiple of 8).  Without one, or for the other
 * cases, one workgroup holds a whole row on chip (vocabulary <= 131072). */
size_t llada_rows_workspace_bytes(int32_t n_rows);
int llada_remask_scores(const llada_remask_args* args, void* stream);

/* llada_remask_scores for callers that cannot allocate device memory or bring a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * arguments and results as llada_remask_scores (bit-identical); args->workspace / workspace_bytes are ignored:
 * the statistics of the streamed row pieces go to a 2 MiB device array built into the library (one copy per
 * GPU, held until the process ends), so the call allocates nothing.  At most LLADA_STATIC_REMASK_MAX_ROWS
 * logits rows per call (LLADA_ERR_WORKSPACE beyond).  The first call on a device looks up the array's address
 * (no allocation; make that call outside a graph capture).  All calls on one GPU share the array: calls that
 * overlap on one GPU (different streams or threads) race; nothing in it needs clearing between calls (every
 * entry a call reads was written by that call). */
#define LLADA_STATIC_REMASK_MAX_ROWS 8192
int llada_remask_scores_static(const llada_remask_args* args, void* stream);

/* generate.py:130-134: per row b, the k = sched[b*sched_stride + step] largest confidences
 * (NaN ranks highest, ties go to the lower index) are transferred: x[b, i] = x0[b, i]. */
int llada_topk_transfer(int64_t* x, const int64_t* x0, const float* confidence, int32_t B, int32_t L,
                        const int32_t* sched, int32_t sched_stride, int32_t step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Likelihood: replaces get_log_likelihood.py:7-26 and :66-75 around the forward.
 * ------------------------------------------------------------------------------------------- */

/* forward_process (get_log_likelihood.py:7-26) for one MC iteration: row r of noisy [b, l]
 * gets counts[r] mask tokens at uniformly random answer positions (columns >= prompt_len),
 * p_mask[r] = counts[r] / target_len is left to the host.  perms [b, target_len] int32 gives the
 * per-row permutations replacing torch.randperm (position j is masked iff perms[r, j] <
 * counts[r]), or NULL = the library's Philox stream (the counts[r] smallest keys are masked). */
int llada_ll_mask(const int64_t* seq, int64_t* noisy, int32_t b, int32_t l, int32_t prompt_len, int64_t mask_id,
                  const int32_t* counts, const int32_t* perms, uint64_t seed, uint32_t iteration,
                  uint32_t row_offset, void* stream);

/* The logits rows get_log_likelihood.py:72 actually reads (`logits[mask_index]`): flat indices
 * r*l + column, ascending, of the mask tokens among the answer columns of noisy [b, l] -> rows_out;
 * row r's entries start at counts[0] + ... + counts[r-1] (counts as passed to llada_ll_mask, so the
 * host knows the total without reading anything back).  rows_out [sum(counts)] int32. */
int llada_ll_masked_rows(const int64_t* noisy, int32_t b, int32_t l, int32_t prompt_len, int64_t mask_id,
                         const int32_t* counts, int32_t* rows_out, void* stream);

/* get_log_likelihood.py:72-73: for logits rows covering the answer positions (row_pos as
 * above), sum over rows whose noisy token is the mask of CE(logits_row, seq)/p_mask[row's
 * sequence] -> *loss_sum (float64, fixed summation order).  CFG combine fused as above.
 * round_ce_bf16 != 0 rounds each CE to bf16 first, as F.cross_entropy on bf16 logits does.
 * row_loss [n_rows] float32 scratch.  workspace (llada_rows_workspace_bytes(n_rows) bytes, 16-byte
 * aligned, or NULL): as for llada_remask_scores, rows are streamed in pieces and merged. */
int llada_masked_ce(const void* logits, const void* uncond_logits, double cfg_scale, const int32_t* row_pos,
                    int32_t n_rows, int32_t vocab, const int64_t* noisy, const int64_t* seq, int32_t b, int32_t l,
                    int64_t mask_id, const float* p_mask_rows, int32_t round_ce_bf16, float* row_loss,
                    double* loss_sum, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LLADA_HIP_H */
