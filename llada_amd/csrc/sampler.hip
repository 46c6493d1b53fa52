// Remasking epilogue of one denoising step (generate.py:81-134 after the forward), HBM-bound.
//
//   remask_rows_kernel : one workgroup per logits row that can matter (a mask token inside the
//                        current block).  Fuses the CFG combine (generate.py:89-90), the EOS/EoT
//                        controls (:94-95, :100-101), add_gumbel_noise in float64 (:8-19),
//                        argmax (:98), the vocab softmax and the confidence rule (:103-123), the
//                        block clamp and the two where() (:125-128).
//   topk_transfer_kernel: per batch row, torch.topk + the masked scatter (:130-134) with k read
//                        from the device schedule, so the step loop never synchronises the host.
//   num_transfer_kernel : get_num_transfer_tokens (:22-40) on the block slice (:75-79).
#include <mutex>

#include "common.h"
#include "philox.h"
#include "rowregs.h"
#include "rowstream.h"

namespace {

struct ArgD {
    double v;
    int i;
};
struct ArgF {
    float v;
    int i;
};

// torch.argmax semantics: first index among equal maxima.
template <class A>
__device__ __forceinline__ A arg_better(A a, A b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <class A>
__device__ __forceinline__ A wave_argmax(A a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        A b;
        b.v = __shfl_xor(a.v, o);
        b.i = __shfl_xor(a.i, o);
        a = arg_better(a, b);
    }
    return a;
}

__global__ void remask_default_kernel(const int64_t* __restrict__ x, int64_t* __restrict__ x0,
                                      float* __restrict__ conf, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x0[i] = x[i];
        conf[i] = -INFINITY;
    }
}

// One 1024-thread workgroup per logits row; the row is read from memory once into registers
// (rowregs.h) and every sweep below runs on them.
template <bool GUMBEL>
__global__ void __launch_bounds__(ROW_THREADS) remask_rows_kernel(const llada_remask_args a) {
    constexpr int NW = ROW_THREADS / 64;
    __shared__ float red_f[3][NW];
    __shared__ int red_i[NW];
    __shared__ double red_d[NW];
    extern __shared__ __attribute__((aligned(16))) unsigned char row_lds[];   // ROW_LDS_BYTES

    const int r = blockIdx.x;
    const int pos = a.row_pos ? a.row_pos[r] : r;
    const int b = pos / a.L, col = pos - b * a.L;
    if (a.x[pos] != a.mask_id || col >= a.block_end) return;  // defaults already written (uniform exit)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.vocab, chunks = V / 8;
    const bf16_t* lc = (const bf16_t*)a.logits + (size_t)r * V;
    const bf16_t* lu = a.uncond_logits ? (const bf16_t*)a.uncond_logits + (size_t)r * V : nullptr;
    const float cfg1 = (float)(a.cfg_scale + 1.0);

    RowRegs row;
    const bool eos_inf = (a.flags & LLADA_FLAG_LOGITS_EOS_INF) != 0;                 // generate.py:94-95
    if (!GUMBEL) {
        row.load(lc, lu, cfg1, chunks, tid, row_lds);
        if (eos_inf) row.exclude(a.eos_id, tid);
    }

    // ---- sweep 1: argmax (float64 Gumbel-max when sampling, generate.py:16-19,98) ----
    int x0;
    float best_logit = 0.f;   // T == 0: the winning logit = the softmax maximum when nothing is excluded
    if (GUMBEL) {
        // The float64 pow/log/exp per entry dominate this sweep and need the registers, so it streams
        // the row from memory; the register copy for the softmax sweeps is loaded afterwards (from
        // the cache hierarchy).
        ArgD bd = {-(double)INFINITY, 0x7fffffff};
        uint32_t k0, k1;
        philox_key(a.seed, LLADA_STREAM_GUMBEL, k0, k1);
        const double temp = (double)a.temperature;
#pragma unroll 1
        for (int c = tid; c < chunks; c += ROW_THREADS) {
            u32x4 qv = *(const u32x4*)(lc + (size_t)c * 8);
            if (lu) {
                const u32x4 uv = *(const u32x4*)(lu + (size_t)c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    qv[e] = pack_bf16x2(cfg_combine(bf16lo(qv[e]), bf16lo(uv[e]), cfg1),
                                        cfg_combine(bf16hi(qv[e]), bf16hi(uv[e]), cfg1));
            }
            double u[8];
            if (a.noise) {
                const double* np = a.noise + (size_t)r * V + (size_t)c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = np[e];
            } else {
#pragma unroll
                for (int h2 = 0; h2 < 4; ++h2) {
                    const Philox4 ph = philox4x32_10((uint32_t)(c * 4 + h2), (uint32_t)col, a.row_offset + (uint32_t)b,
                                                     a.step, k0, k1);
                    u[2 * h2] = u53(ph.v[0], ph.v[1]);
                    u[2 * h2 + 1] = u53(ph.v[2], ph.v[3]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int v = c * 8 + e;
                float l = (e & 1) ? bf16hi(qv[e >> 1]) : bf16lo(qv[e >> 1]);
                if (eos_inf && v == a.eos_id) l = -INFINITY;
                // generate.py:16-19 ranks exp(logits) / (-log u) ** T in float64; its logarithm
                // l - T * log(-log u) has the same arg-max and needs two logarithms instead of a
                // pow, an exp and a division (this sweep is bound by float64 arithmetic)
                const double sc = (double)l - temp * log(-log(u[e]));
                if (sc > bd.v || (sc == bd.v && v < bd.i)) {
                    bd.v = sc;
                    bd.i = v;
                }
            }
        }
        bd = wave_argmax(bd);
        if (lane == 0) {
            red_d[wave] = bd.v;
            red_i[wave] = bd.i;
        }
        __syncthreads();
        bd.v = red_d[0];
        bd.i = red_i[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) bd = arg_better(bd, ArgD{red_d[w], red_i[w]});
        x0 = bd.i;
        row.load(lc, lu, cfg1, chunks, tid, row_lds);
        if (eos_inf) row.exclude(a.eos_id, tid);
    } else {
        ArgF bf = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int i = 0; i < ROW_SLOTS; ++i) {
            const int v0 = (tid + i * ROW_THREADS) * 8;   // ascending within a lane: '>' keeps the first maximum
            const u32x4 t = row.get(i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float l = (e & 1) ? bf16hi(t[e >> 1]) : bf16lo(t[e >> 1]);
                if (l > bf.v) {
                    bf.v = l;
                    bf.i = v0 + e;
                }
            }
        }
        if (bf.i == 0x7fffffff) bf.i = tid * 8;           // a lane that saw only -inf: its first entry
        bf = wave_argmax(bf);
        if (lane == 0) {
            red_f[0][wave] = bf.v;
            red_i[wave] = bf.i;
        }
        __syncthreads();
        bf.v = red_f[0][0];
        bf.i = red_i[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) bf = arg_better(bf, ArgF{red_f[0][w], red_i[w]});
        x0 = bf.i;
        best_logit = bf.v;
    }

    float conf;
    if (a.remasking == LLADA_REMASK_RANDOM) {
        if (a.rand_scores) {
            conf = a.rand_scores[pos];
        } else {
            uint32_t r0, r1;
            philox_key(a.seed, LLADA_STREAM_REMASK, r0, r1);
            const Philox4 ph = philox4x32_10((uint32_t)col, 0u, a.row_offset + (uint32_t)b, a.step, r0, r1);
            conf = (float)(ph.v[0] >> 8) * (1.0f / 16777216.0f);
        }
    } else {
        // generate.py:101 `logits_with_noise[..EOS] = logits[..EoT] = -inf` after x0 is taken: at
        // temperature 0 both names are one tensor, so the softmax loses EOS and EoT; otherwise only EoT.
        int excl0 = -1, excl1 = -1;
        if (a.flags & LLADA_FLAG_CONF_EOS_EOT_INF) {
            excl0 = a.eot_id;
            if (!GUMBEL) excl1 = a.eos_id;
            row.exclude(excl0, tid);
            row.exclude(excl1, tid);
        }
        // ---- sweep 2: softmax maximum (known already at temperature 0 without exclusions) ----
        float m;
        if (!GUMBEL && excl0 < 0)
            m = best_logit;
        else
            m = block_max16(row.max(), red_f[1], lane, wave);
        // ---- sweep 3: sum of exp(l - max) ----
        const float s = block_sum16(row.sum_exp(m), red_f[2], lane, wave);
        if (a.remasking == LLADA_REMASK_LOW_CONFIDENCE || a.remasking == LLADA_REMASK_HIGH_CONFIDENCE) {
            // p[x0], rounded to bf16 like F.softmax on a bf16 tensor
            float lx = bf16_to_f32(lc[x0]);
            if (lu) lx = cfg_combine(lx, bf16_to_f32(lu[x0]), cfg1);
            if (((a.flags & LLADA_FLAG_LOGITS_EOS_INF) && x0 == a.eos_id) || x0 == excl0 || x0 == excl1) lx = -INFINITY;
            const bool f32 = (a.flags & LLADA_FLAG_CONF_FLOAT32) != 0;   // eval_llada.py:187 softmaxes in float32
            const float p = f32 ? expf(lx - m) / s : round_bf16(expf(lx - m) / s);
            conf = (a.remasking == LLADA_REMASK_LOW_CONFIDENCE) ? p : (f32 ? 1.0f - p : round_bf16(1.0f - p));
        } else {
            // ---- sweep 4: sum of bf16(log_p * p), both factors rounded to bf16 first ----
            const float lse = logf(s), inv_s = 1.0f / s;
            float esum = 0.f;
#pragma unroll
            for (int i = 0; i < ROW_SLOTS; ++i) {
                if (tid + i * ROW_THREADS >= chunks) continue;
                const u32x4 t = row.get(i);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float l = (e & 1) ? bf16hi(t[e >> 1]) : bf16lo(t[e >> 1]);
                    const float z = l - m;
                    const float p = round_bf16(__expf(z) * inv_s);
                    const float lp = round_bf16(z - lse);
                    esum += round_bf16(lp * p);
                }
            }
            __syncthreads();   // red_f[1] may still be read by the maximum above
            const float tot = round_bf16(block_sum16(esum, red_f[1], lane, wave));
            conf = (a.remasking == LLADA_REMASK_LOW_ENTROPY) ? -tot : tot;
        }
    }
    if (tid == 0) {
        a.x0[pos] = (int64_t)x0;
        a.confidence[pos] = conf;
    }
}


// ---- temperature 0, confidence rules and 'random': streamed statistics of row pieces (rowstream.h) ----
// Block b < n_stream: piece b % splits of logits row b / splits.  The remaining blocks write the
// defaults (x0 = x, confidence = -inf) of every position; the merge kernel, next on the stream,
// overwrites the active rows.  ex0 / ex1: vocabulary entries the softmax does not see (or -1).
template <bool NEED_SUM>
__global__ void __launch_bounds__(RS_THREADS) remask_stream_kernel(const llada_remask_args a, int splits, int cps,
                                                                   int ex0, int ex1, RowPart* __restrict__ parts,
                                                                   int n_stream) {
    __shared__ float red_f[2 * RS_THREADS / 64];
    __shared__ int red_i[RS_THREADS / 64];
    if ((int)blockIdx.x >= n_stream) {
        const int64_t n = (int64_t)a.B * a.L;
        const int64_t stride = (int64_t)(gridDim.x - n_stream) * RS_THREADS;
        for (int64_t i = (int64_t)(blockIdx.x - n_stream) * RS_THREADS + threadIdx.x; i < n; i += stride) {
            a.x0[i] = a.x[i];
            a.confidence[i] = -INFINITY;
        }
        return;
    }
    const int r = blockIdx.x / splits, sp = blockIdx.x - r * splits;
    const int pos = a.row_pos ? a.row_pos[r] : r;
    const int col = pos % a.L;
    if (a.x[pos] != a.mask_id || col >= a.block_end) return;      // uniform exit: the defaults stand
    const int chunks = a.vocab / 8;
    const int c_begin = sp * cps, c_end = min(chunks, c_begin + cps);
    const bf16_t* lc = (const bf16_t*)a.logits + (size_t)r * a.vocab;
    const bf16_t* lu = a.uncond_logits ? (const bf16_t*)a.uncond_logits + (size_t)r * a.vocab : nullptr;
    const RowPart p = row_stream_part<NEED_SUM, true>(lc, lu, (float)(a.cfg_scale + 1.0), ex0, ex1, c_begin, c_end,
                                                      red_f, red_i);
    if (threadIdx.x == 0) parts[(size_t)r * splits + sp] = p;
}

// Temperature > 0: the float64 Gumbel-max of generate.py:16-19,98 over one piece of a row (same ranking
// as remask_rows_kernel<true>), after the piece's softmax statistics.  The piece is streamed twice (the
// second time from the cache hierarchy); the float64 logarithms dominate either way.  With 32 or 128
// active rows (one prompt) the pieces spread that arithmetic over eight times as many CUs.
struct GumbelPart {
    double v;
    int i;
    int pad;
};

// Float32 evaluation of the Gumbel-max score l - T log(-log u), u a float64 uniform in [0, 1).
// w = -log u: for u < 1/2, -logf((float)u): (float)u carries 2^-24 relative error, i.e. 6e-8 absolute in the
// logarithm, logf adds at most one ulp of a result of magnitude <= 37 (4e-6), and w >= 0.69, so w is good to about
// 2^-22 relative; for u >= 1/2, 1 - u is exact in float64 (Sterbenz), its rounding to float32 is 2^-24 relative and
// -log1pf(-(1 - u)) a couple of ulp: again about 2^-22 relative, however close u is to 1.  g = logf(w) then carries that
// relative error as absolute error plus one ulp of |g| <= 37 (4e-6); T * g and the final subtraction round once each
// (ulp of a magnitude below 128: 8e-6).  Together under 2.5e-5 (1 + T); PRESCREEN_ERR = 2.5e-4 leaves a factor of ten.
// u == 0 gives -inf, like the float64 expression.  No contraction, no fast-math in this build: the value is a pure
// function of its arguments.
constexpr float PRESCREEN_ERR = 2.5e-4f;
__device__ __forceinline__ float gumbel_score32(float l, float T, double u) {
    float w;
    if (u < 0.5) {
        w = -logf((float)u);
    } else {
        w = -log1pf(-(float)(1.0 - u));
    }
    return l - T * logf(w);
}

__global__ void __launch_bounds__(RS_THREADS) remask_stream_gumbel_kernel(const llada_remask_args a, int splits, int cps,
                                                                          int ex0, int ex1, RowPart* __restrict__ parts,
                                                                          GumbelPart* __restrict__ gparts, int n_stream) {
    constexpr int NWV = RS_THREADS / 64;
    __shared__ float red_f[2 * NWV];
    __shared__ int red_i[NWV];
    __shared__ double red_d[NWV];
    if ((int)blockIdx.x >= n_stream) {
        const int64_t n = (int64_t)a.B * a.L;
        const int64_t stride = (int64_t)(gridDim.x - n_stream) * RS_THREADS;
        for (int64_t i = (int64_t)(blockIdx.x - n_stream) * RS_THREADS + threadIdx.x; i < n; i += stride) {
            a.x0[i] = a.x[i];
            a.confidence[i] = -INFINITY;
        }
        return;
    }
    const int r = blockIdx.x / splits, sp = blockIdx.x - r * splits;
    const int pos = a.row_pos ? a.row_pos[r] : r;
    const int b = pos / a.L, col = pos - b * a.L;
    if (a.x[pos] != a.mask_id || col >= a.block_end) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.vocab, chunks = V / 8;
    const int c_begin = sp * cps, c_end = min(chunks, c_begin + cps);
    const bf16_t* lc = (const bf16_t*)a.logits + (size_t)r * V;
    const bf16_t* lu = a.uncond_logits ? (const bf16_t*)a.uncond_logits + (size_t)r * V : nullptr;
    const float cfg1 = (float)(a.cfg_scale + 1.0);
    const bool need_sum = a.remasking != LLADA_REMASK_RANDOM;
    RowPart p = {-INFINITY, 0.f, 0, 0};
    if (need_sum) p = row_stream_part<true, false>(lc, lu, cfg1, ex0, ex1, c_begin, c_end, red_f, red_i);

    const bool eos_inf = (a.flags & LLADA_FLAG_LOGITS_EOS_INF) != 0;
    ArgD bd = {-(double)INFINITY, 0x7fffffff};
    uint32_t k0, k1;
    philox_key(a.seed, LLADA_STREAM_GUMBEL, k0, k1);
    const double temp = (double)a.temperature;
    // Float32 pre-screen.  The float64 score l - T log(-log u) decides the arg-max, and its two float64 logarithms are
    // what this sweep costs; a float32 evaluation of the same expression is within PRESCREEN_ERR (1 + T) of it for every
    // u in [0, 1) (-log u through log1p(-(1 - u)) above one half, where 1 - u is exact in float64; error budget in the
    // comment of gumbel_score32), so an entry whose float32 score lies more than twice that bound under the largest
    // float32 score the wave has seen -- this trip included -- is strictly worse in float64 than the entry that set it
    // and cannot be the arg-max, nor tie with it.  Only the others (a handful per piece) are ranked in float64, with
    // the same comparison and the same first-index rule as before: the result is the full float64 sweep's, bit for bit.
    const float tf = a.temperature;
    const float slack = 2.0f * PRESCREEN_ERR * (1.0f + fabsf(tf));
    float wave_max = -INFINITY;
#pragma unroll 1
    for (int cb = c_begin; cb < c_end; cb += RS_THREADS) {          // the same trips for every lane: the wave maximum needs all
        const int c = cb + tid;
        const bool active = c < c_end;
        const int cl = active ? c : c_end - 1;
        const u32x4 qv = rs_chunk(lc, lu, cfg1, cl, -1, -1);
        double u[8];
        if (a.noise) {
            const double* np = a.noise + (size_t)r * V + (size_t)cl * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = np[e];
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) {
                const Philox4 ph = philox4x32_10((uint32_t)(cl * 4 + h2), (uint32_t)col, a.row_offset + (uint32_t)b, a.step,
                                                 k0, k1);
                u[2 * h2] = u53(ph.v[0], ph.v[1]);
                u[2 * h2 + 1] = u53(ph.v[2], ph.v[3]);
            }
        }
        float l[8], s32[8];
        float lane_max = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = cl * 8 + e;
            l[e] = (e & 1) ? bf16hi(qv[e >> 1]) : bf16lo(qv[e >> 1]);
            if (eos_inf && v == a.eos_id) l[e] = -INFINITY;
            s32[e] = active ? gumbel_score32(l[e], tf, u[e]) : -INFINITY;
            lane_max = fmaxf(lane_max, s32[e]);                     // fmaxf drops a NaN score: such an entry never wins below either
        }
        wave_max = fmaxf(wave_max, wave_reduce_max(lane_max));
        const float th = wave_max - slack;                          // -inf - slack = -inf: nothing is skipped while all scores are -inf
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (active && s32[e] >= th) {
                const int v = cl * 8 + e;
                const double sc = (double)l[e] - temp * log(-log(u[e]));    // same arg-max as exp(l) / (-log u) ** T
                if (sc > bd.v || (sc == bd.v && v < bd.i)) {
                    bd.v = sc;
                    bd.i = v;
                }
            }
        }
    }
    bd = wave_argmax(bd);
    __syncthreads();             // red_i may still be read by the statistics above
    if (lane == 0) {
        red_d[wave] = bd.v;
        red_i[wave] = bd.i;
    }
    __syncthreads();
    if (tid == 0) {
        bd.v = red_d[0];
        bd.i = red_i[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) bd = arg_better(bd, ArgD{red_d[w], red_i[w]});
        parts[(size_t)r * splits + sp] = p;
        GumbelPart gp;
        gp.v = bd.v;
        gp.i = bd.i;
        gp.pad = 0;
        gparts[(size_t)r * splits + sp] = gp;
    }
}

// One thread per logits row: merges the pieces in index order, then generate.py:98-128 on the merged
// statistics.  xa0 / xa1: entries that compete for the arg-max although the softmax does not see them
// (generate.py:100-101 removes EOS / EoT after x0 is taken), or -1.
__global__ void __launch_bounds__(256) remask_merge_kernel(const llada_remask_args a, int splits, int ex0, int ex1,
                                                           int xa0, int xa1, const RowPart* __restrict__ parts,
                                                           const GumbelPart* __restrict__ gparts) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    const int pos = a.row_pos ? a.row_pos[r] : r;
    const int b = pos / a.L, col = pos - b * a.L;
    // the pieces are fetched before the activity test so that their loads travel with that of x[pos]
    // (an inactive row's stale pieces are read and dropped)
    const RowPart* p = parts + (size_t)r * splits;
    RowPart pk[RS_MAX_SPLITS];
#pragma unroll
    for (int k = 0; k < RS_MAX_SPLITS; ++k) pk[k] = p[k < splits ? k : 0];
    if (a.x[pos] != a.mask_id || col >= a.block_end) return;
    float M = pk[0].m, S = pk[0].s;
    int x0 = pk[0].idx;
#pragma unroll
    for (int k = 1; k < RS_MAX_SPLITS; ++k) {
        if (k >= splits) break;
        if (pk[k].m > M) x0 = pk[k].idx;    // strict: the first piece keeps a repeated maximum
        rs_merge(M, S, pk[k].m, pk[k].s);
    }
    const bf16_t* lc = (const bf16_t*)a.logits + (size_t)r * a.vocab;
    const bf16_t* lu = a.uncond_logits ? (const bf16_t*)a.uncond_logits + (size_t)r * a.vocab : nullptr;
    const float cfg1 = (float)(a.cfg_scale + 1.0);
    auto logit = [&](int v) {
        float l = bf16_to_f32(lc[v]);
        if (lu) l = cfg_combine(l, bf16_to_f32(lu[v]), cfg1);
        return l;
    };
    if (gparts) {
        // temperature > 0: x0 is the Gumbel arg-max; its logit is read back (it need not be the maximum)
        ArgD bd = {gparts[(size_t)r * splits].v, gparts[(size_t)r * splits].i};
        for (int k = 1; k < splits; ++k) bd = arg_better(bd, ArgD{gparts[(size_t)r * splits + k].v, gparts[(size_t)r * splits + k].i});
        x0 = bd.i;
        float conf;
        if (a.remasking == LLADA_REMASK_RANDOM) {
            if (a.rand_scores) {
                conf = a.rand_scores[pos];
            } else {
                uint32_t r0, r1;
                philox_key(a.seed, LLADA_STREAM_REMASK, r0, r1);
                const Philox4 ph = philox4x32_10((uint32_t)col, 0u, a.row_offset + (uint32_t)b, a.step, r0, r1);
                conf = (float)(ph.v[0] >> 8) * (1.0f / 16777216.0f);
            }
        } else {
            float lx = logit(x0);
            if (x0 == ex0 || x0 == ex1) lx = -INFINITY;
            const bool f32 = (a.flags & LLADA_FLAG_CONF_FLOAT32) != 0;
            const float pr = f32 ? expf(lx - M) / S : round_bf16(expf(lx - M) / S);
            conf = (a.remasking == LLADA_REMASK_LOW_CONFIDENCE) ? pr : (f32 ? 1.0f - pr : round_bf16(1.0f - pr));
        }
        a.x0[pos] = (int64_t)x0;
        a.confidence[pos] = conf;
        return;
    }
    float best = M;
    bool seen = true;            // x0 is an entry the softmax sees: its logit is the maximum M
    const int xa[2] = {xa0, xa1};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (xa[i] < 0) continue;
        const float l = logit(xa[i]);
        if (l > best || (l == best && xa[i] < x0)) {
            best = l;
            x0 = xa[i];
            seen = false;
        }
    }
    float conf;
    if (a.remasking == LLADA_REMASK_RANDOM) {
        if (a.rand_scores) {
            conf = a.rand_scores[pos];
        } else {
            uint32_t r0, r1;
            philox_key(a.seed, LLADA_STREAM_REMASK, r0, r1);
            const Philox4 ph = philox4x32_10((uint32_t)col, 0u, a.row_offset + (uint32_t)b, a.step, r0, r1);
            conf = (float)(ph.v[0] >> 8) * (1.0f / 16777216.0f);
        }
    } else {
        const float lx = seen ? M : -INFINITY;
        const bool f32 = (a.flags & LLADA_FLAG_CONF_FLOAT32) != 0;
        const float pr = f32 ? expf(lx - M) / S : round_bf16(expf(lx - M) / S);
        conf = (a.remasking == LLADA_REMASK_LOW_CONFIDENCE) ? pr : (f32 ? 1.0f - pr : round_bf16(1.0f - pr));
    }
    a.x0[pos] = (int64_t)x0;
    a.confidence[pos] = conf;
}


// Order-preserving key of a confidence: NaN ranks highest (torch.topk), -0 == +0.
__device__ __forceinline__ uint32_t conf_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256) topk_transfer_kernel(int64_t* __restrict__ x, const int64_t* __restrict__ x0,
                                                            const float* __restrict__ conf, int L,
                                                            const int32_t* __restrict__ sched, int sched_stride,
                                                            int step) {
    extern __shared__ uint32_t keys[];  // [L]
    __shared__ int red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int k = sched[(size_t)b * sched_stride + step];
    if (k <= 0) return;
    if (k > L) k = L;
    const uint32_t KEY_NEG_INF = conf_key(-INFINITY);
    int nc = 0;
    for (int i = tid; i < L; i += 256) {
        const uint32_t key = conf_key(conf[(size_t)b * L + i]);
        keys[i] = key;
        nc += key > KEY_NEG_INF;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nc += __shfl_xor(nc, o);
    if ((tid & 63) == 0) red[tid >> 6] = nc;
    __syncthreads();
    nc = red[0] + red[1] + red[2] + red[3];
    const bool only_candidates = k <= nc;  // then no -inf entry can be among the k largest
    for (int i = tid; i < L; i += 256) {
        const uint32_t ki = keys[i];
        if (only_candidates && ki <= KEY_NEG_INF) continue;
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            const uint32_t kj = keys[j];
            rank += (kj > ki) || (kj == ki && j < i);
        }
        if (rank < k) x[(size_t)b * L + i] = x0[(size_t)b * L + i];
    }
}

__global__ void __launch_bounds__(256) num_transfer_kernel(const int64_t* __restrict__ x, int L, int block_start,
                                                           int block_end, int64_t mask_id, int steps,
                                                           int32_t* __restrict__ sched) {
    __shared__ int red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = 0;
    for (int i = block_start + tid; i < block_end; i += 256) n += x[(size_t)b * L + i] == mask_id;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((tid & 63) == 0) red[tid >> 6] = n;
    __syncthreads();
    n = red[0] + red[1] + red[2] + red[3];
    const int base = n / steps, rem = n % steps;
    for (int s = tid; s < steps; s += 256) sched[(size_t)b * steps + s] = base + (s < rem ? 1 : 0);
}

// add_gumbel_noise of generate.py:8-19 as a stand-alone map (the sampler itself never materialises it: the
// fused epilogue ranks l - T*log(-log u) in registers): out = exp(l) / (-log u)^T in float64.  One thread per
// 8 consecutive entries, the same Philox counters as the fused kernels, so argmax(out) is the epilogue's x0.
__global__ void __launch_bounds__(256) gumbel_noise_kernel(const double* __restrict__ logits, const double* __restrict__ noise,
                                                           double temp, double* __restrict__ out, int64_t n_rows, int V,
                                                           int L, uint64_t seed, uint32_t step, uint32_t row_offset) {
    const int chunks = V / 8;
    uint32_t k0, k1;
    philox_key(seed, LLADA_STREAM_GUMBEL, k0, k1);
    const int64_t total = n_rows * chunks;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        const uint32_t b = (uint32_t)(r / L), col = (uint32_t)(r % L);
        const size_t base = (size_t)r * V + (size_t)c * 8;
        double u[8];
        if (noise) {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = noise[base + e];
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) {
                const Philox4 ph = philox4x32_10((uint32_t)(c * 4 + h2), col, row_offset + b, step, k0, k1);
                u[2 * h2] = u53(ph.v[0], ph.v[1]);
                u[2 * h2 + 1] = u53(ph.v[2], ph.v[3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) out[base + e] = exp(logits[base + e]) / pow(-log(u[e]), temp);   // generate.py:17-19
    }
}

}  // namespace

extern "C" int llada_add_gumbel_noise(const double* logits, const double* noise, double temperature, double* out,
                                      int64_t n_rows, int32_t vocab, int32_t L, uint64_t seed, uint32_t step,
                                      uint32_t row_offset, void* stream) {
    LLADA_CHECK_ARG(logits && out && n_rows > 0 && vocab > 0 && L > 0, "add_gumbel_noise: bad argument");
    LLADA_CHECK_SUPPORTED(vocab % 8 == 0, "add_gumbel_noise: vocab=%d must be a multiple of 8", vocab);
    const int64_t total = n_rows * (vocab / 8);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(gumbel_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, noise, temperature, out,
                       n_rows, vocab, L, seed, step, row_offset);
    LLADA_CHECK_LAUNCH("gumbel_noise_kernel");
    return LLADA_OK;
}

extern "C" int llada_num_transfer_tokens(const int64_t* x, int32_t B, int32_t L, int32_t block_start,
                                         int32_t block_end, int64_t mask_id, int32_t steps, int32_t* sched,
                                         void* stream) {
    LLADA_CHECK_ARG(x && sched && B > 0 && L > 0 && steps > 0, "num_transfer_tokens: bad argument");
    LLADA_CHECK_ARG(0 <= block_start && block_start <= block_end && block_end <= L, "num_transfer_tokens: bad block");
    hipLaunchKernelGGL(num_transfer_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, L, block_start, block_end,
                       mask_id, steps, sched);
    LLADA_CHECK_LAUNCH("num_transfer_kernel");
    return LLADA_OK;
}

extern "C" int llada_remask_scores(const llada_remask_args* args, void* stream) {
    LLADA_CHECK_ARG(args != nullptr, "remask_scores: null args");
    const llada_remask_args& a = *args;
    LLADA_CHECK_ARG(a.logits && a.x && a.x0 && a.confidence, "remask_scores: null tensor");
    LLADA_CHECK_ARG(a.n_rows > 0 && a.B > 0 && a.L > 0 && a.vocab > 0, "remask_scores: bad shape");
    LLADA_CHECK_SUPPORTED(a.vocab % 8 == 0, "remask_scores: vocab=%d must be a multiple of 8", a.vocab);
    LLADA_CHECK_ARG(a.row_pos != nullptr || a.n_rows == a.B * a.L, "remask_scores: row_pos is required unless n_rows == B*L");
    if (a.remasking < LLADA_REMASK_LOW_CONFIDENCE || a.remasking > LLADA_REMASK_RANDOM) {
        llada_set_error("remask_scores: unknown remasking rule %d", a.remasking);
        return LLADA_ERR_UNSUPPORTED;  // NotImplementedError(remasking), generate.py:122-123
    }
    if (a.flags & (LLADA_FLAG_LOGITS_EOS_INF | LLADA_FLAG_CONF_EOS_EOT_INF))
        LLADA_CHECK_ARG(a.eos_id >= 0 && a.eos_id < a.vocab && a.eot_id >= 0 && a.eot_id < a.vocab,
                        "remask_scores: eos/eot id outside the vocabulary");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)a.B * a.L;
    const int g = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    const bool entropy = a.remasking == LLADA_REMASK_LOW_ENTROPY || a.remasking == LLADA_REMASK_HIGH_ENTROPY;
    if (!entropy && a.workspace != nullptr) {
        // streamed row pieces + merge: needs only (max, first arg-max / Gumbel arg-max, sum of exponentials) per row
        LLADA_CHECK_ARG(((uintptr_t)a.workspace & 15) == 0, "remask_scores: workspace must be 16-byte aligned");
        const bool gumbel = a.temperature != 0.0f;
        const int chunks = a.vocab / 8;
        // the float64 Gumbel sweep is arithmetic-bound: always the finest cut that keeps the chip busy
        int splits = rs_pick_splits(a.n_rows, chunks, llada_rows_forced_splits);
        if (gumbel && llada_rows_forced_splits <= 0) {
            splits = 1;
            while (splits < RS_MAX_SPLITS && (int64_t)a.n_rows * splits < 2048 && chunks / (2 * splits) >= RS_THREADS) splits *= 2;
        }
        const size_t per_piece = sizeof(RowPart) + sizeof(GumbelPart);
        if ((size_t)a.n_rows * splits * per_piece > a.workspace_bytes) {
            llada_set_error("remask_scores: workspace of %zu bytes, %zu needed (llada_rows_workspace_bytes)",
                            a.workspace_bytes, (size_t)a.n_rows * splits * per_piece);
            return LLADA_ERR_WORKSPACE;
        }
        LLADA_CHECK_SUPPORTED((int64_t)a.n_rows * splits + g < 0x7fffffffLL, "remask_scores: grid too large");
        const int cps = (chunks + splits - 1) / splits;
        const bool le = (a.flags & LLADA_FLAG_LOGITS_EOS_INF) != 0, ce = (a.flags & LLADA_FLAG_CONF_EOS_EOT_INF) != 0;
        RowPart* parts = (RowPart*)a.workspace;
        GumbelPart* gparts = (GumbelPart*)(parts + (size_t)a.n_rows * splits);
        const int n_stream = a.n_rows * splits;
        if (gumbel) {
            // generate.py:94-95 removes EOS everywhere; :100-101 removes only EoT from the softmax (the noisy
            // copy that loses EOS is not used again)
            const int ex0 = le ? a.eos_id : -1, ex1 = ce ? a.eot_id : -1;
            hipLaunchKernelGGL(remask_stream_gumbel_kernel, dim3(n_stream + g), dim3(RS_THREADS), 0, s, a, splits, cps, ex0,
                               ex1, parts, gparts, n_stream);
            LLADA_CHECK_LAUNCH("remask_stream_gumbel_kernel");
            hipLaunchKernelGGL(remask_merge_kernel, dim3((a.n_rows + 255) / 256), dim3(256), 0, s, a, splits, ex0, ex1, -1, -1,
                               (const RowPart*)parts, (const GumbelPart*)gparts);
            LLADA_CHECK_LAUNCH("remask_merge_kernel");
            return LLADA_OK;
        }
        // generate.py:94-95 removes EOS before the arg-max; :100-101 (temperature 0: one tensor under two
        // names) removes EOS and EoT from the softmax only
        const int ex0 = (le || ce) ? a.eos_id : -1, ex1 = ce ? a.eot_id : -1;
        const int xa0 = (ce && !le) ? a.eos_id : -1, xa1 = ce ? a.eot_id : -1;
        if (a.remasking == LLADA_REMASK_RANDOM)
            hipLaunchKernelGGL(remask_stream_kernel<false>, dim3(n_stream + g), dim3(RS_THREADS), 0, s, a, splits, cps, ex0,
                               ex1, parts, n_stream);
        else
            hipLaunchKernelGGL(remask_stream_kernel<true>, dim3(n_stream + g), dim3(RS_THREADS), 0, s, a, splits, cps, ex0,
                               ex1, parts, n_stream);
        LLADA_CHECK_LAUNCH("remask_stream_kernel");
        hipLaunchKernelGGL(remask_merge_kernel, dim3((a.n_rows + 255) / 256), dim3(256), 0, s, a, splits, ex0, ex1, xa0, xa1,
                           (const RowPart*)parts, (const GumbelPart*)nullptr);
        LLADA_CHECK_LAUNCH("remask_merge_kernel");
        return LLADA_OK;
    }
    LLADA_CHECK_SUPPORTED(a.vocab <= ROW_MAX_VOCAB, "remask_scores: vocab=%d exceeds %d (one row must fit the registers of a workgroup)", a.vocab, ROW_MAX_VOCAB);
    hipLaunchKernelGGL(remask_default_kernel, dim3(g), dim3(256), 0, s, a.x, a.x0, a.confidence, n);
    LLADA_CHECK_LAUNCH("remask_default_kernel");
    static OncePerDevice attr_set;
    if (attr_set.first()) {
        hipError_t e1 = hipFuncSetAttribute((const void*)remask_rows_kernel<true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, ROW_LDS_BYTES);
        hipError_t e2 = hipFuncSetAttribute((const void*)remask_rows_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, ROW_LDS_BYTES);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            llada_set_error("hipFuncSetAttribute(remask_rows_kernel): %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return LLADA_ERR_HIP;
        }
    }
    if (a.temperature != 0.0f)
        hipLaunchKernelGGL(remask_rows_kernel<true>, dim3(a.n_rows), dim3(ROW_THREADS), ROW_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL(remask_rows_kernel<false>, dim3(a.n_rows), dim3(ROW_THREADS), ROW_LDS_BYTES, s, a);
    LLADA_CHECK_LAUNCH("remask_rows_kernel");
    return LLADA_OK;
}

// ---- llada_remask_scores without any allocation: the piece statistics live in this module's own device array ----
// Room for LLADA_STATIC_REMASK_MAX_ROWS logits rows cut into RS_MAX_SPLITS pieces (2 MiB; every GPU the library
// runs on holds its own copy until the process ends).  Host code must not use the array's name as an address (it is a
// host-side stand-in): the device address is looked up once per device and only that pointer is handed to kernels.
constexpr size_t SCRATCH_BYTES = (size_t)LLADA_STATIC_REMASK_MAX_ROWS * RS_MAX_SPLITS * (sizeof(RowPart) + sizeof(GumbelPart));
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_lock;
void* scratch_ptr[MAX_DEVICES] = {};   // device address of `scratch` on each device, resolved on first use

// Device address of this device's copy of `scratch` (hipGetSymbolAddress allocates nothing).
int scratch_address(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        llada_set_error("remask_scores_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return LLADA_ERR_HIP;
    }
    std::lock_guard<std::mutex> guard(scratch_lock);
    if (scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            llada_set_error("remask_scores_static: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
            return LLADA_ERR_HIP;
        }
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return LLADA_OK;
}
}  // namespace

extern "C" int llada_remask_scores_static(const llada_remask_args* args, void* stream) {
    LLADA_CHECK_ARG(args != nullptr, "remask_scores_static: null args");
    if (args->n_rows > LLADA_STATIC_REMASK_MAX_ROWS) {
        llada_set_error("remask_scores_static: n_rows=%d exceeds the built-in scratch of %d rows (use llada_remask_scores "
                        "with a workspace)", args->n_rows, LLADA_STATIC_REMASK_MAX_ROWS);
        return LLADA_ERR_WORKSPACE;
    }
    void* p = nullptr;
    const int rc = scratch_address(&p);
    if (rc != LLADA_OK) return rc;
    llada_remask_args a = *args;
    a.workspace = p;                  // the looked-up device address, never the array's name
    a.workspace_bytes = SCRATCH_BYTES;
    return llada_remask_scores(&a, stream);
}

int llada_rows_forced_splits = 0;   // test hook: 0 = automatic
extern "C" void llada_rows_force_splits(int n) { llada_rows_forced_splits = n; }

extern "C" size_t llada_rows_workspace_bytes(int32_t n_rows) {
    return n_rows > 0 ? (size_t)n_rows * RS_MAX_SPLITS * (sizeof(RowPart) + sizeof(GumbelPart)) : 0;
}

extern "C" int llada_topk_transfer(int64_t* x, const int64_t* x0, const float* confidence, int32_t B, int32_t L,
                                   const int32_t* sched, int32_t sched_stride, int32_t step, void* stream) {
    LLADA_CHECK_ARG(x && x0 && confidence && sched && B > 0 && L > 0, "topk_transfer: bad argument");
    LLADA_CHECK_ARG(step >= 0 && step < sched_stride, "topk_transfer: step outside the schedule");
    LLADA_CHECK_SUPPORTED(L <= 15360, "topk_transfer: L=%d too long", L);
    hipLaunchKernelGGL(topk_transfer_kernel, dim3(B), dim3(256), (size_t)L * sizeof(uint32_t), (hipStream_t)stream, x,
                       x0, confidence, L, sched, sched_stride, step);
    LLADA_CHECK_LAUNCH("topk_transfer_kernel");
    return LLADA_OK;
}
