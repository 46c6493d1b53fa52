// The non-causal LLaDA-8B forward as one C call: replaces `model(x, attention_mask=...).logits`
// (generate.py:88,92; get_log_likelihood.py:38).  Host-side orchestration only: it enqueues the
// kernels of gemm.hip / attention.hip / elementwise.hip on the caller's stream and returns; all
// scratch lives in the caller's workspace.  The LM head runs only on the rows the caller asks
// for: the sampler needs logits just for the current block (everything else is overwritten by
// generate.py:125-128), the likelihood just for the answer positions.
#include "common.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t x, h, qkv, att, act, gu, hf, splitk, splitk_bytes, total;
};

Layout make_layout(const llada_config& c, int64_t M, int64_t R) {
    Layout o{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align256(bytes);
        return at;
    };
    const size_t d = c.d_model, F = c.mlp_hidden;
    o.x = take(M * d * 2);
    o.h = take(M * d * 2);
    o.qkv = take(M * 3 * d * 2);
    o.att = take(M * d * 2);
    o.act = take(M * F * 2);
    o.gu = take(M * 2 * F * 2);
    o.hf = take(R * d * 2);
    // split-K scratch for the small-M regime (every projection of a layer, and the LM head)
    size_t sk = 0;
    const size_t shapes[8][3] = {{(size_t)M, 3 * d, d}, {(size_t)M, d, d}, {(size_t)M, 2 * F, d}, {(size_t)M, d, F},
                                 {(size_t)R, (size_t)c.vocab_size, d},
                                 {(size_t)R, d, d}, {(size_t)R, 2 * F, d}, {(size_t)R, d, F}};   // last layer, wanted rows only
    for (auto& sh : shapes) {
        const size_t b = llada_gemm_workspace_bytes((int64_t)sh[0], (int32_t)sh[1], (int32_t)sh[2]);
        sk = b > sk ? b : sk;
    }
    o.splitk_bytes = sk;
    o.splitk = take(sk);
    o.total = off;
    return o;
}

}  // namespace

extern "C" size_t llada_forward_workspace_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len,
                                                int32_t n_logit_rows) {
    if (!cfg || n_seq <= 0 || seq_len <= 0 || n_logit_rows <= 0) return 0;
    return make_layout(*cfg, (int64_t)n_seq * seq_len, n_logit_rows).total;
}

#define TRY(call)                    \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != LLADA_OK) return rc_; \
    } while (0)

namespace {

// Copies the K | V thirds of the compact rows of one block (qkv_b [n_seq * block_len, 3 d]) into the rows
// seq * seq_len + block_start + i of a per-layer [Q | K | V] image.
__global__ void __launch_bounds__(256) scatter_kv_kernel(const bf16_t* __restrict__ qkv_b, bf16_t* __restrict__ image,
                                                         int64_t n_rows, int block_len, int seq_len, int block_start,
                                                         int d) {
    const int chunks = 2 * d / 8;   // 16-byte pieces of K | V per row
    const int64_t total = n_rows * chunks;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        const int64_t dst_row = (r / block_len) * seq_len + block_start + (r % block_len);
        *(u32x4*)(image + dst_row * 3 * d + d + c * 8) = *(const u32x4*)(qkv_b + r * 3 * d + d + c * 8);
    }
}

int forward_impl(const llada_config* cfg, const llada_weights* w, const int64_t* ids, const uint8_t* key_mask,
                 const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                 const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out, void* workspace,
                 size_t workspace_bytes, void* kv_cache, const int32_t* positions, void* stream);

}  // namespace

extern "C" int llada_forward_positions(const llada_config* cfg, const llada_weights* w, const int64_t* ids,
                                       const uint8_t* key_mask, const int32_t* positions, const float* rope_cos,
                                       const float* rope_sin, int32_t n_seq, int32_t seq_len, const int32_t* logit_rows,
                                       int32_t n_logit_rows, void* logits_out, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    return forward_impl(cfg, w, ids, key_mask, rope_cos, rope_sin, n_seq, seq_len, logit_rows, n_logit_rows, logits_out,
                        workspace, workspace_bytes, nullptr, positions, stream);
}

extern "C" int llada_forward(const llada_config* cfg, const llada_weights* w, const int64_t* ids,
                             const uint8_t* key_mask, const float* rope_cos, const float* rope_sin, int32_t n_seq,
                             int32_t seq_len, const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(cfg, w, ids, key_mask, rope_cos, rope_sin, n_seq, seq_len, logit_rows, n_logit_rows, logits_out,
                        workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" size_t llada_kv_cache_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len) {
    if (!cfg || n_seq <= 0 || seq_len <= 0) return 0;
    return (size_t)cfg->n_layers * n_seq * seq_len * 3 * cfg->d_model * 2;
}

extern "C" int llada_forward_fill_cache(const llada_config* cfg, const llada_weights* w, const int64_t* ids,
                                        const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                                        int32_t n_seq, int32_t seq_len, const int32_t* logit_rows, int32_t n_logit_rows,
                                        void* logits_out, void* workspace, size_t workspace_bytes, void* kv_cache,
                                        void* stream) {
    LLADA_CHECK_ARG(kv_cache != nullptr && ((uintptr_t)kv_cache & 255) == 0, "forward_fill_cache: kv_cache must be 256-byte aligned");
    return forward_impl(cfg, w, ids, key_mask, rope_cos, rope_sin, n_seq, seq_len, logit_rows, n_logit_rows, logits_out,
                        workspace, workspace_bytes, kv_cache, nullptr, stream);
}

namespace {

int forward_impl(const llada_config* cfg, const llada_weights* w, const int64_t* ids, const uint8_t* key_mask,
                 const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                 const int32_t* logit_rows, int32_t n_logit_rows, void* logits_out, void* workspace,
                 size_t workspace_bytes, void* kv_cache, const int32_t* positions, void* stream) {
    LLADA_CHECK_ARG(cfg && w && ids && rope_cos && rope_sin && logits_out && workspace, "forward: null argument");
    LLADA_CHECK_ARG(n_seq > 0 && seq_len > 0 && n_logit_rows > 0, "forward: bad shape");
    const llada_config& c = *cfg;
    LLADA_CHECK_SUPPORTED(c.n_heads > 0 && c.d_model == c.n_heads * 128, "forward: head_dim must be 128");
    LLADA_CHECK_SUPPORTED(c.d_model % 256 == 0 && c.mlp_hidden % 64 == 0 && c.vocab_size % 8 == 0,
                          "forward: d_model %% 256, mlp_hidden %% 64, vocab %% 8 must be 0");
    const int64_t M = (int64_t)n_seq * seq_len;
    LLADA_CHECK_ARG(logit_rows != nullptr || n_logit_rows == M, "forward: logit_rows is required unless all rows are wanted");
    const Layout lay = make_layout(c, M, n_logit_rows);
    if (workspace_bytes < lay.total) {
        llada_set_error("forward: workspace has %zu bytes, %zu needed", workspace_bytes, lay.total);
        return LLADA_ERR_WORKSPACE;
    }
    unsigned char* ws = (unsigned char*)workspace;
    void* x = ws + lay.x;
    void* h = ws + lay.h;
    void* const qkv_ws = ws + lay.qkv;
    void* qkv = qkv_ws;
    void* att = ws + lay.att;
    void* act = ws + lay.act;
    void* gu = ws + lay.gu;
    void* hf = ws + lay.hf;
    void* sk = lay.splitk_bytes ? ws + lay.splitk : nullptr;
    const size_t skb = lay.splitk_bytes;
    const int d = c.d_model, F = c.mlp_hidden, V = c.vocab_size;
    const int tl = (w->packed & 2) ? 1 : 0;            // projection matrices stored tile-major
    const int tf = tl ? LLADA_GEMM_W_TILED : 0;

    // In the last layer only the wanted rows matter once attention has mixed the tokens: they are
    // compacted (attention output and residual stream) and the output projection, the MLP and the
    // final norm run on those rows alone.  Scratch: the qkv buffer is dead after attention.
    const bool prune = logit_rows != nullptr && (int64_t)n_logit_rows * 2 <= M;
    const int64_t R = n_logit_rows;
    void* xc = qkv_ws;                                     // [R, d] compacted residual stream
    void* hc = (unsigned char*)qkv_ws + (size_t)R * d * 2; // [R, d] its norm

    TRY(llada_embed_bf16(w->wte, ids, x, M, d, V, stream));
    // Every residual projection is issued together with the norm that reads its result (the output
    // projection with ff_norm, the down projection with the next layer's attn_norm or the final norm):
    // llada_gemm_residual_rmsnorm_bf16 folds the norm into the GEMM's second pass when it has one.
    if (c.n_layers > 0) TRY(llada_rmsnorm_bf16(x, w->attn_norm[0], h, M, d, c.rms_eps, nullptr, stream));
    bool final_norm_done = false;
    for (int l = 0; l < c.n_layers; ++l) {
        const bool last = l == c.n_layers - 1;
        // with a cache every layer leaves its [Q | K | V] image there instead of in the shared scratch
        if (kv_cache) qkv = (unsigned char*)kv_cache + (size_t)l * M * 3 * d * 2;
        if (w->packed & 1) {
            TRY(llada_gemm_bf16_positions(h, w->wqkv[l], qkv, M, 3 * d, d, d, 3 * d, LLADA_EPI_ROPE_QKV | tf, nullptr, 0,
                                          rope_cos, rope_sin, seq_len, positions, sk, skb, stream));
        } else {
            TRY(llada_gemm_bf16(h, w->wqkv[l], qkv, M, 3 * d, d, d, 3 * d, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr,
                                nullptr, 0, sk, skb, stream));
            TRY(llada_rope_bf16_positions(qkv, rope_cos, rope_sin, n_seq, seq_len, c.n_heads, positions, stream));
        }
        TRY(llada_attn_bidir_bf16(qkv, key_mask, att, n_seq, seq_len, c.n_heads, stream));
        int64_t Ml = M;      // rows from here to the end of the layer
        void* xl = x;        // residual stream of those rows
        void* al = att;      // attention output of those rows
        void* hl = h;
        if (prune && last) {
            TRY(llada_gather_rows_bf16(att, logit_rows, h, R, d, M, stream));   // h is free until the second norm
            TRY(llada_gather_rows_bf16(x, logit_rows, xc, R, d, M, stream));
            Ml = R;
            xl = xc;
            al = h;
            hl = hc;
        }
        TRY(llada_gemm_residual_rmsnorm_bf16(al, w->wo[l], xl, hl, w->ff_norm[l], c.rms_eps, Ml, d, d, d, tl, sk, skb, stream));
        if (w->packed & 1) {
            TRY(llada_gemm_bf16(hl, w->wgu[l], act, Ml, 2 * F, d, d, F, LLADA_EPI_SWIGLU | tf, nullptr, 0, nullptr, nullptr,
                                0, sk, skb, stream));
        } else {
            TRY(llada_gemm_bf16(hl, w->wgu[l], gu, Ml, 2 * F, d, d, 2 * F, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr,
                                nullptr, 0, sk, skb, stream));
            TRY(llada_swiglu_bf16(gu, act, Ml, F, stream));
        }
        if (!last) {
            TRY(llada_gemm_residual_rmsnorm_bf16(act, w->wdown[l], x, h, w->attn_norm[l + 1], c.rms_eps, M, d, F, F, tl, sk,
                                                 skb, stream));
        } else if (prune) {   // the compacted rows are exactly the rows of the final norm
            TRY(llada_gemm_residual_rmsnorm_bf16(act, w->wdown[l], xl, hf, w->ln_f, c.rms_eps, Ml, d, F, F, tl, sk, skb,
                                                 stream));
            final_norm_done = true;
        } else {
            TRY(llada_gemm_bf16(act, w->wdown[l], xl, Ml, d, F, F, d, LLADA_EPI_RESIDUAL | tf, xl, d, nullptr, nullptr, 0,
                                sk, skb, stream));
        }
    }
    if (!final_norm_done) {
        if (prune && c.n_layers > 0)
            TRY(llada_rmsnorm_bf16(xc, w->ln_f, hf, R, d, c.rms_eps, nullptr, stream));
        else
            TRY(llada_rmsnorm_bf16(x, w->ln_f, hf, R, d, c.rms_eps, logit_rows, stream));
    }
    TRY(llada_gemm_bf16(hf, w->lm_head, logits_out, R, V, d, d, V, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr,
                        nullptr, 0, sk, skb, stream));
    return LLADA_OK;
}

}  // namespace

// One denoising step of the block-cached sampler: only the block_len positions starting at block_start of
// every sequence are recomputed (block_ids [n_seq, block_len]); their keys and values replace the cached ones
// and their queries attend over the whole cached sequence.  logits_out [n_seq * block_len, vocab].
extern "C" int llada_forward_block(const llada_config* cfg, const llada_weights* w, const int64_t* block_ids,
                                   const uint8_t* key_mask, const float* rope_cos, const float* rope_sin, int32_t n_seq,
                                   int32_t seq_len, int32_t block_start, int32_t block_len, void* kv_cache,
                                   int32_t update_cache, const int32_t* logit_rows, int32_t n_logit_rows,
                                   void* logits_out, void* workspace, size_t workspace_bytes, void* stream) {
    LLADA_CHECK_ARG(cfg && w && block_ids && rope_cos && rope_sin && logits_out && workspace && kv_cache,
                    "forward_block: null argument");
    LLADA_CHECK_ARG(logit_rows == nullptr || n_logit_rows > 0, "forward_block: logit_rows without a count");
    LLADA_CHECK_ARG(n_seq > 0 && seq_len > 0 && block_len > 0 && block_start >= 0 && block_start + block_len <= seq_len,
                    "forward_block: bad shape");
    const llada_config& c = *cfg;
    LLADA_CHECK_SUPPORTED(c.n_heads > 0 && c.d_model == c.n_heads * 128, "forward: head_dim must be 128");
    LLADA_CHECK_SUPPORTED(c.d_model % 256 == 0 && c.mlp_hidden % 64 == 0 && c.vocab_size % 8 == 0,
                          "forward: d_model %% 256, mlp_hidden %% 64, vocab %% 8 must be 0");
    const int64_t Mb = (int64_t)n_seq * block_len;
    const int64_t M = (int64_t)n_seq * seq_len;
    const Layout lay = make_layout(c, Mb, Mb);
    const size_t attn_bytes = llada_attn_workspace_bytes(n_seq, block_len, seq_len, c.n_heads);   // key-split partials
    if (workspace_bytes < lay.total + attn_bytes) {
        llada_set_error("forward_block: workspace has %zu bytes, %zu needed", workspace_bytes, lay.total + attn_bytes);
        return LLADA_ERR_WORKSPACE;
    }
    unsigned char* ws = (unsigned char*)workspace;
    void* attn_ws = attn_bytes ? ws + lay.total : nullptr;
    void* x = ws + lay.x;
    void* h = ws + lay.h;
    void* qkv = ws + lay.qkv;
    void* att = ws + lay.att;
    void* act = ws + lay.act;
    void* gu = ws + lay.gu;
    void* hf = ws + lay.hf;
    void* sk = lay.splitk_bytes ? ws + lay.splitk : nullptr;
    const size_t skb = lay.splitk_bytes;
    const int d = c.d_model, F = c.mlp_hidden, V = c.vocab_size;
    const int tl = (w->packed & 2) ? 1 : 0;            // projection matrices stored tile-major
    const int tf = tl ? LLADA_GEMM_W_TILED : 0;
    // rotary tables of the block's positions: row block_start onwards, period block_len
    const float* cos_b = rope_cos + (size_t)block_start * 64;
    const float* sin_b = rope_sin + (size_t)block_start * 64;
    hipStream_t st = (hipStream_t)stream;

    TRY(llada_embed_bf16(w->wte, block_ids, x, Mb, d, V, stream));
    if (c.n_layers > 0) TRY(llada_rmsnorm_bf16(x, w->attn_norm[0], h, Mb, d, c.rms_eps, nullptr, stream));
    for (int l = 0; l < c.n_layers; ++l) {
        const bool last = l == c.n_layers - 1;
        bf16_t* image = (bf16_t*)((unsigned char*)kv_cache + (size_t)l * M * 3 * d * 2);
        if (w->packed & 1) {
            TRY(llada_gemm_bf16(h, w->wqkv[l], qkv, Mb, 3 * d, d, d, 3 * d, LLADA_EPI_ROPE_QKV | tf, nullptr, 0, cos_b, sin_b,
                                block_len, sk, skb, stream));
        } else {
            TRY(llada_gemm_bf16(h, w->wqkv[l], qkv, Mb, 3 * d, d, d, 3 * d, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr, nullptr,
                                0, sk, skb, stream));
            TRY(llada_rope_bf16(qkv, cos_b, sin_b, n_seq, block_len, c.n_heads, stream));
        }
        if (update_cache) {
            const int64_t total = Mb * (2 * d / 8);
            const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            hipLaunchKernelGGL(scatter_kv_kernel, dim3(grid), dim3(256), 0, st, (const bf16_t*)qkv, image, Mb, block_len,
                               seq_len, block_start, d);
            LLADA_CHECK_LAUNCH("scatter_kv_kernel");
        }
        TRY(llada_attn_bidir_queries_bf16(qkv, (int64_t)3 * d, block_len, image, key_mask, att, n_seq, seq_len, c.n_heads,
                                          update_cache ? nullptr : qkv, block_start, attn_ws, attn_bytes, stream));
        TRY(llada_gemm_residual_rmsnorm_bf16(att, w->wo[l], x, h, w->ff_norm[l], c.rms_eps, Mb, d, d, d, tl, sk, skb, stream));
        if (w->packed & 1) {
            TRY(llada_gemm_bf16(h, w->wgu[l], act, Mb, 2 * F, d, d, F, LLADA_EPI_SWIGLU | tf, nullptr, 0, nullptr, nullptr, 0,
                                sk, skb, stream));
        } else {
            TRY(llada_gemm_bf16(h, w->wgu[l], gu, Mb, 2 * F, d, d, 2 * F, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr, nullptr,
                                0, sk, skb, stream));
            TRY(llada_swiglu_bf16(gu, act, Mb, F, stream));
        }
        TRY(llada_gemm_residual_rmsnorm_bf16(act, w->wdown[l], x, last ? hf : h, last ? w->ln_f : w->attn_norm[l + 1],
                                             c.rms_eps, Mb, d, F, F, tl, sk, skb, stream));
    }
    if (c.n_layers == 0) TRY(llada_rmsnorm_bf16(x, w->ln_f, hf, Mb, d, c.rms_eps, nullptr, stream));
    if (logit_rows) {   // LM head on the wanted rows only: their normalised rows are compacted into h (free by now)
        TRY(llada_gather_rows_bf16(hf, logit_rows, h, n_logit_rows, d, Mb, stream));
        TRY(llada_gemm_bf16(h, w->lm_head, logits_out, n_logit_rows, V, d, d, V, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr,
                            nullptr, 0, sk, skb, stream));
        return LLADA_OK;
    }
    TRY(llada_gemm_bf16(hf, w->lm_head, logits_out, Mb, V, d, d, V, LLADA_EPI_PLAIN | tf, nullptr, 0, nullptr, nullptr, 0, sk,
                        skb, stream));
    return LLADA_OK;
}

extern "C" size_t llada_forward_block_workspace_bytes(const llada_config* cfg, int32_t n_seq, int32_t seq_len,
                                                      int32_t block_len) {
    if (!cfg || n_seq <= 0 || block_len <= 0 || seq_len <= 0) return 0;
    const int64_t Mb = (int64_t)n_seq * block_len;
    return make_layout(*cfg, Mb, Mb).total + llada_attn_workspace_bytes(n_seq, block_len, seq_len, cfg->n_heads);
}
