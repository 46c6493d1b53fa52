// HBM-bound pieces of the mask predictor's forward (embedding lookup, RMSNorm, rotary embedding,
// SwiGLU), all 16-byte vectorised.  They replace the ATen elementwise/normalisation kernels of
// the model the reference calls at generate.py:88,92 / get_log_likelihood.py:38.
#include "common.h"

namespace {

// ---- embedding: out[m, :] = wte[ids[m], :] ------------------------------------------------
__global__ void embed_kernel(const bf16_t* __restrict__ wte, const int64_t* __restrict__ ids, bf16_t* __restrict__ out,
                             int64_t n_tokens, int d, int64_t vocab) {
    const int chunks = d / 8;
    for (int64_t m = blockIdx.x; m < n_tokens; m += gridDim.x) {
        int64_t id = ids[m];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // memory safety only: the Python host (check_token_ids) rejects such ids before the call
        const u32x4* src = (const u32x4*)(wte + id * d);
        u32x4* dst = (u32x4*)(out + m * d);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) dst[c] = src[c];
    }
}

// ---- row gather: out[r, :] = in[rows[r], :] (compacts the rows the LM head needs) ----------------
__global__ void gather_rows_kernel(const bf16_t* __restrict__ in, const int32_t* __restrict__ rows, bf16_t* __restrict__ out,
                                   int64_t n_rows, int d, int64_t n_in) {
    const int chunks = d / 8;
    for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        int64_t src_row = rows[r];
        src_row = src_row < 0 ? 0 : (src_row >= n_in ? n_in - 1 : src_row);  // memory safety only: the row lists are built by the library's own host code
        const u32x4* src = (const u32x4*)(in + src_row * d);
        u32x4* dst = (u32x4*)(out + r * d);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) dst[c] = src[c];
    }
}

// ---- RMSNorm ---------------------------------------------------------------------------------
// One 256-thread workgroup per row; d <= 256*8*MAXV elements are kept in registers between the
// sum-of-squares pass and the scaling pass.
template <int MAXV>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int64_t n_rows, int d, float eps,
                                                      const int32_t* __restrict__ row_index) {
    __shared__ float red[4];
    const int chunks = d / 8;
    const int tid = threadIdx.x;
    for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int64_t src_row = row_index ? (int64_t)row_index[r] : r;
        const u16x8* xp = (const u16x8*)(x + src_row * d);
        u16x8 v[MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + i * 256;
            if (c < chunks) {
                v[i] = xp[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf16_to_f32(v[i][e]);
                    ss += f * f;
                }
            }
        }
        ss = wave_reduce_sum(ss);
        __syncthreads();  // red[] reuse across rows
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float rs = rsqrtf(tot / (float)d + eps);
        const u16x8* wp = (const u16x8*)w;
        u16x8* op = (u16x8*)(out + r * d);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + i * 256;
            if (c < chunks) {
                const u16x8 wv = wp[c];
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xn = round_bf16(bf16_to_f32(v[i][e]) * rs);
                    o[e] = f32_to_bf16(xn * bf16_to_f32(wv[e]));
                }
                op[c] = o;
            }
        }
    }
}

// The same norm with one WAVE per row (four rows per 256-thread workgroup, d = 256 * 8 * MAXV exactly): a lane keeps the
// chunks of the four threads tid = lane + 64 w of the kernel above, adds their squares in the same order, reduces each
// of the four partial sums with the same butterfly and adds them as red[0..3] are added there, so every output bit
// is the same; what changes is that a row's 8 * MAXV loads per lane are all in flight at once and no workgroup barrier
// sits between a row's two passes.
template <int MAXV>
__global__ void __launch_bounds__(256) rmsnorm_wave_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ out, int64_t n_rows, float eps,
                                                           const int32_t* __restrict__ row_index) {
    constexpr int d = 256 * 8 * MAXV;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int64_t src_row = row_index ? (int64_t)row_index[r] : r;
    const u16x8* xp = (const u16x8*)(x + src_row * d);
    u16x8 v[4][MAXV];
#pragma unroll
    for (int wv = 0; wv < 4; ++wv)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) v[wv][i] = xp[lane + 64 * wv + 256 * i];
    float tot = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf16_to_f32(v[wv][i][e]);
                ss += f * f;
            }
        ss = wave_reduce_sum(ss);
        tot = wv == 0 ? ss : tot + ss;
    }
    const float rs = rsqrtf(tot / (float)d + eps);
    const u16x8* wp = (const u16x8*)w;
    u16x8* op = (u16x8*)(out + r * d);
#pragma unroll
    for (int wv = 0; wv < 4; ++wv)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * wv + 256 * i;
            const u16x8 wvv = wp[c];
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xn = round_bf16(bf16_to_f32(v[wv][i][e]) * rs);
                o[e] = f32_to_bf16(xn * bf16_to_f32(wvv[e]));
            }
            op[c] = o;
        }
}

// ---- rotary embedding, natural head layout (pairs j, j+64), in place on Q and K -----------------
__global__ void rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cs, const float* __restrict__ sn,
                            int64_t n_tokens, int L, int H, const int* __restrict__ positions) {
    // one thread per (token, q/k, head, 8 frequencies)
    const int64_t total = n_tokens * 2 * H * 8;
    const int d = H * 128;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 7);
        const int h = (int)((i >> 3) % H);
        const int qk = (int)((i / (8 * (int64_t)H)) & 1);
        const int64_t m = i / (16 * (int64_t)H);
        const int pos = positions ? min(max(positions[m], 0), L - 1) : (int)(m % L);
        bf16_t* p = qkv + m * 3 * d + (size_t)qk * d + h * 128 + c * 8;
        const u16x8 a = *(const u16x8*)p;
        const u16x8 b = *(const u16x8*)(p + 64);
        const float* cp = cs + (size_t)pos * 64 + c * 8;
        const float* sp = sn + (size_t)pos * 64 + c * 8;
        u16x8 oa, ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x1 = bf16_to_f32(a[e]), x2 = bf16_to_f32(b[e]);
            oa[e] = f32_to_bf16(x1 * cp[e] - x2 * sp[e]);
            ob[e] = f32_to_bf16(x2 * cp[e] + x1 * sp[e]);
        }
        *(u16x8*)p = oa;
        *(u16x8*)(p + 64) = ob;
    }
}

// ---- SwiGLU, natural layout gu = [gate | up] -------------------------------------------------
__global__ void swiglu_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, int64_t n_rows, int F) {
    const int chunks = F / 8;
    const int64_t total = n_rows * chunks;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / chunks;
        const int c = (int)(i % chunks);
        const u16x8 g = *(const u16x8*)(gu + m * 2 * F + c * 8);
        const u16x8 u = *(const u16x8*)(gu + m * 2 * F + F + c * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = bf16_to_f32(g[e]);
            const float act = round_bf16(silu_f32(gf));
            o[e] = f32_to_bf16(act * bf16_to_f32(u[e]));
        }
        *(u16x8*)(out + m * F + c * 8) = o;
    }
}

int g_rmsnorm_wave = 1;   // test hook: 0 = the one-workgroup-per-row kernel at every width

}  // namespace

extern "C" void llada_rmsnorm_use_wave_kernel(int on) { g_rmsnorm_wave = on; }

extern "C" int llada_gather_rows_bf16(const void* in, const int32_t* rows, void* out, int64_t n_rows, int32_t d,
                                      int64_t n_in, void* stream) {
    LLADA_CHECK_ARG(in && rows && out && n_rows > 0 && n_in > 0, "gather_rows: bad argument");
    LLADA_CHECK_SUPPORTED(d % 8 == 0, "gather_rows: d=%d must be a multiple of 8", d);
    const int grid = (int)(n_rows < 4096 ? n_rows : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, rows,
                       (bf16_t*)out, n_rows, d, n_in);
    LLADA_CHECK_LAUNCH("gather_rows_kernel");
    return LLADA_OK;
}

extern "C" int llada_embed_bf16(const void* wte, const int64_t* ids, void* out, int64_t n_tokens, int32_t d,
                                int64_t vocab, void* stream) {
    LLADA_CHECK_ARG(wte && ids && out && n_tokens > 0 && vocab > 0, "embed: bad argument");
    LLADA_CHECK_SUPPORTED(d % 8 == 0, "embed: d=%d must be a multiple of 8", d);
    const int grid = (int)(n_tokens < 4096 ? n_tokens : 4096);
    hipLaunchKernelGGL(embed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wte, ids,
                       (bf16_t*)out, n_tokens, d, vocab);
    LLADA_CHECK_LAUNCH("embed_kernel");
    return LLADA_OK;
}

extern "C" int llada_rmsnorm_bf16(const void* x, const void* w, void* out, int64_t n_rows, int32_t d, float eps,
                                  const int32_t* row_index, void* stream) {
    LLADA_CHECK_ARG(x && w && out && n_rows > 0, "rmsnorm: bad argument");
    LLADA_CHECK_SUPPORTED(d % 8 == 0 && d <= 256 * 8 * 4, "rmsnorm: d=%d must be a multiple of 8 and <= 8192", d);
    const int grid = (int)(n_rows < 8192 ? n_rows : 8192);
    if (g_rmsnorm_wave && d == 256 * 8 * 2 && n_rows < (1ll << 31)) {      // the LLaDA-8B width: one wave per row
        hipLaunchKernelGGL(rmsnorm_wave_kernel<2>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out, n_rows, eps, row_index);
        LLADA_CHECK_LAUNCH("rmsnorm_wave_kernel");
        return LLADA_OK;
    }
    if (d <= 256 * 8 * 2)
        hipLaunchKernelGGL(rmsnorm_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)out, n_rows, d, eps, row_index);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)out, n_rows, d, eps, row_index);
    LLADA_CHECK_LAUNCH("rmsnorm_kernel");
    return LLADA_OK;
}

extern "C" int llada_rope_bf16(void* qkv, const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                               int32_t n_heads, void* stream) {
    return llada_rope_bf16_positions(qkv, rope_cos, rope_sin, n_seq, seq_len, n_heads, nullptr, stream);
}

int llada_rope_bf16_positions(void* qkv, const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                              int32_t n_heads, const int32_t* positions, void* stream) {
    LLADA_CHECK_ARG(qkv && rope_cos && rope_sin && n_seq > 0 && seq_len > 0 && n_heads > 0, "rope: bad argument");
    const int64_t n_tokens = (int64_t)n_seq * seq_len;
    const int64_t total = n_tokens * 16 * n_heads;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, rope_cos, rope_sin,
                       n_tokens, seq_len, n_heads, positions);
    LLADA_CHECK_LAUNCH("rope_kernel");
    return LLADA_OK;
}

extern "C" int llada_swiglu_bf16(const void* gu, void* out, int64_t n_rows, int32_t F, void* stream) {
    LLADA_CHECK_ARG(gu && out && n_rows > 0 && F > 0, "swiglu: bad argument");
    LLADA_CHECK_SUPPORTED(F % 8 == 0, "swiglu: F=%d must be a multiple of 8", F);
    const int64_t total = n_rows * (F / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(swiglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (bf16_t*)out,
                       n_rows, F);
    LLADA_CHECK_LAUNCH("swiglu_kernel");
    return LLADA_OK;
}
