// Streaming statistics of one piece of a full-vocabulary logits row: maximum, first arg-max and sum of
// exponentials in ONE pass, nothing kept on chip.
//
// The temperature-0 confidence rules of the sampler epilogue (generate.py:98,103-109) and the masked
// cross-entropy (get_log_likelihood.py:72) need only those three numbers of a row, and they merge
// associatively, so a row can be cut into up to RS_MAX_SPLITS contiguous pieces, one 256-thread
// workgroup each: the few rows of a denoising step (8 prompts x block_length 32 = 256, or 32 for one
// prompt) then still put about two workgroups on every CU, and with thousands of rows the small
// workgroups of different rows overlap their loads and their arithmetic (a whole-row workgroup loads
// first and computes afterwards, alone on its CU).  A lane reads RS_UNROLL 16-byte chunks per trip
// (lane-strided: every load instruction of a wave covers 1 KiB of consecutive addresses), keeps a
// running (maximum, sum) pair rescaled once per trip, and remembers only WHICH trip first held its
// maximum; the few lanes whose maximum is the piece's maximum then look for the first index in that
// trip -- in registers when it was the last trip (always, for a piece of one trip), re-read otherwise.
// A tiny second kernel merges the pieces (split order = index order, so "first maximum" and the
// summation order are fixed: results are deterministic).
#pragma once
#include "common.h"
#include "rowregs.h"

constexpr int RS_THREADS = 256;
constexpr int RS_UNROLL = 8;       // 16-byte chunks in flight per lane
constexpr int RS_MAX_SPLITS = 8;   // a full-vocabulary row / 8 = 7.7 chunks per lane = one trip
constexpr float RS_LOG2E = 1.4426950408889634f;

typedef __attribute__((ext_vector_type(2))) float f32x2;

extern int llada_rows_forced_splits;   // test hook (sampler.hip): 0 = automatic

struct RowPart {
    float m;    // maximum of the piece (entries excluded from the softmax count as -inf)
    float s;    // sum of exp(l - m) over the piece (0 when m == -inf)
    int idx;    // vocabulary index of the first maximum
    int pad;
};

// max(a, b, c) in one instruction (the logits are bit patterns to the compiler, so fmaxf() would
// first canonicalise every operand with a v_max of its own)
__device__ __forceinline__ float rs_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float rs_safe(float m) { return m == -INFINITY ? 0.f : m; }

// (m, s) <- merge of (m, s) and (m2, s2)
__device__ __forceinline__ void rs_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2), mu = rs_safe(mn);
    s = s * __builtin_amdgcn_exp2f((m - mu) * RS_LOG2E) + s2 * __builtin_amdgcn_exp2f((m2 - mu) * RS_LOG2E);
    m = mn;
}

// entry e (0..7) of a chunk -> -inf
__device__ __forceinline__ u32x4 rs_patch(u32x4 t, int e) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w != (e >> 1)) continue;
        t[w] = (e & 1) ? ((t[w] & 0x0000FFFFu) | 0xFF800000u) : ((t[w] & 0xFFFF0000u) | 0x0000FF80u);
    }
    return t;
}

// One chunk (8 logits) as it enters the statistics: combined with the unconditional row under
// classifier-free guidance, vocabulary entries ex0 / ex1 (or -1) replaced by -inf.
__device__ __forceinline__ u32x4 rs_chunk(const bf16_t* lc, const bf16_t* lu, float cfg1, int c, int ex0, int ex1) {
    u32x4 t = *(const u32x4*)(lc + (size_t)c * 8);
    if (lu) {
        const u32x4 b = *(const u32x4*)(lu + (size_t)c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            t[e] = pack_bf16x2(cfg_combine(bf16lo(t[e]), bf16lo(b[e]), cfg1), cfg_combine(bf16hi(t[e]), bf16hi(b[e]), cfg1));
    }
    if (ex0 >= 0 && (ex0 >> 3) == c) t = rs_patch(t, ex0 & 7);
    if (ex1 >= 0 && (ex1 >> 3) == c) t = rs_patch(t, ex1 & 7);
    return t;
}

// Statistics of chunks [c_begin, c_end) of a row by one RS_THREADS workgroup; the result is valid in
// thread 0.  red_f [2 * RS_THREADS / 64] and red_i [RS_THREADS / 64] are LDS scratch.
template <bool NEED_SUM, bool NEED_ARG>
__device__ __forceinline__ RowPart row_stream_part(const bf16_t* lc, const bf16_t* lu, float cfg1, int ex0, int ex1,
                                                   int c_begin, int c_end, float* red_f, int* red_i) {
    constexpr int NWV = RS_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int exc0 = ex0 >= 0 ? (ex0 >> 3) : -1, exc1 = ex1 >= 0 ? (ex1 >> 3) : -1;
    const u32x4 ninf = {BF16_NEG_INF2, BF16_NEG_INF2, BF16_NEG_INF2, BF16_NEG_INF2};
    float m = -INFINITY, s = 0.f;
    int first_trip = 0;
    int trip = 0;
    u32x4 t[RS_UNROLL];             // outlives the loop: the last trip is searched in registers
    for (int c0 = c_begin; c0 < c_end; c0 += RS_THREADS * RS_UNROLL, ++trip) {
#pragma unroll
        for (int j = 0; j < RS_UNROLL; ++j) {
            // unconditional loads (address clamped into the piece) travel together
            const int c = min(c0 + j * RS_THREADS + tid, c_end - 1);
            t[j] = __builtin_nontemporal_load((const u32x4*)(lc + (size_t)c * 8));
        }
        if (lu) {
#pragma unroll
            for (int j = 0; j < RS_UNROLL; ++j) {
                const int c = min(c0 + j * RS_THREADS + tid, c_end - 1);
                const u32x4 b = __builtin_nontemporal_load((const u32x4*)(lu + (size_t)c * 8));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    t[j][e] = pack_bf16x2(cfg_combine(bf16lo(t[j][e]), bf16lo(b[e]), cfg1),
                                          cfg_combine(bf16hi(t[j][e]), bf16hi(b[e]), cfg1));
            }
        }
#pragma unroll
        for (int j = 0; j < RS_UNROLL; ++j) {
            const int c = c0 + j * RS_THREADS + tid;
            if (c >= c_end) t[j] = ninf;
            if (c == exc0) t[j] = rs_patch(t[j], ex0 & 7);
            if (c == exc1) t[j] = rs_patch(t[j], ex1 & 7);
        }
        float bm = -INFINITY;
#pragma unroll
        for (int j = 0; j < RS_UNROLL; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) bm = rs_max3(bm, bf16lo(t[j][w]), bf16hi(t[j][w]));
        if (bm > m) first_trip = trip;      // strict: the earliest trip keeps a repeated maximum
        const float mn = fmaxf(m, bm);
        if constexpr (NEED_SUM) {
            const float mu = rs_safe(mn);
            const f32x2 k2 = {-mu * RS_LOG2E, -mu * RS_LOG2E};
            const f32x2 l2e = {RS_LOG2E, RS_LOG2E};
            f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < RS_UNROLL; ++j)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x2 v = {bf16lo(t[j][w]), bf16hi(t[j][w])};
                    const f32x2 z = __builtin_elementwise_fma(v, l2e, k2);
                    const f32x2 p = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
                    acc[w & 1] += p;
                }
            const f32x2 a = acc[0] + acc[1];
            s = s * __builtin_amdgcn_exp2f((m - mu) * RS_LOG2E) + (a[0] + a[1]);
        }
        m = mn;
    }
    const float m_lane = m;
    // ---- workgroup: fixed-order merge (xor butterflies inside a wave, wave order across) ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        if constexpr (NEED_SUM) {
            // both partners must add in the same order to stay bit-identical across the wave
            const bool low = (lane & o) == 0;
            float ma = low ? m : m2, sa = low ? s : s2;
            const float mb = low ? m2 : m, sb = low ? s2 : s;
            rs_merge(ma, sa, mb, sb);
            m = ma;
            s = sa;
        } else {
            m = fmaxf(m, m2);
        }
    }
    if (lane == 0) {
        red_f[wave] = m;
        red_f[NWV + wave] = s;
    }
    __syncthreads();
    float M = red_f[0], S = red_f[NWV];
#pragma unroll
    for (int w = 1; w < NWV; ++w) {
        if constexpr (NEED_SUM)
            rs_merge(M, S, red_f[w], red_f[NWV + w]);
        else
            M = fmaxf(M, red_f[w]);
    }
    int idx = c_begin * 8;          // a piece of nothing but -inf: its first entry
    if constexpr (NEED_ARG) {
        int mine = 0x7fffffff;
        if (m_lane == M && M != -INFINITY) {
            const int c0 = c_begin + first_trip * (RS_THREADS * RS_UNROLL);
            if (first_trip == trip - 1) {
                // the maximum first showed up in the last trip, whose chunks are still in registers (always
                // the case for a piece of one trip); slots past the end hold -inf and cannot match
#pragma unroll
                for (int j = RS_UNROLL - 1; j >= 0; --j)
#pragma unroll
                    for (int e = 7; e >= 0; --e) {
                        const float l = (e & 1) ? bf16hi(t[j][e >> 1]) : bf16lo(t[j][e >> 1]);
                        if (l == M) mine = (c0 + j * RS_THREADS + tid) * 8 + e;
                    }
            } else {
#pragma unroll 1
                for (int j = 0; j < RS_UNROLL && mine == 0x7fffffff; ++j) {
                    const int c = c0 + j * RS_THREADS + tid;
                    if (c >= c_end) break;
                    const u32x4 tc = rs_chunk(lc, lu, cfg1, c, ex0, ex1);
#pragma unroll
                    for (int e = 7; e >= 0; --e) {
                        const float l = (e & 1) ? bf16hi(tc[e >> 1]) : bf16lo(tc[e >> 1]);
                        if (l == M) mine = c * 8 + e;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o));
        if (lane == 0) red_i[wave] = mine;
        __syncthreads();
        int best = red_i[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) best = min(best, red_i[w]);
        if (M != -INFINITY) idx = best;
    }
    RowPart p;
    p.m = M;
    p.s = S;
    p.idx = idx;
    p.pad = 0;
    return p;
}

// Splits of a launch over n_rows rows of `chunks` 16-byte chunks: about two workgroups per CU (measured at
// the full vocabulary: 256 rows are fastest in 2 pieces, 128 in 4, 32 in 8; from 512 rows on whole rows
// stream at the same rate as pieces and every extra workgroup only adds its fixed cost), no piece shorter
// than half a trip of a workgroup.
inline int rs_pick_splits(int64_t n_rows, int chunks, int forced) {
    int s = 1;
    if (forced > 0) {
        s = forced;
    } else {
        while (s < RS_MAX_SPLITS && n_rows * s * 2 <= 512 && (chunks + 2 * s - 1) / (2 * s) >= RS_THREADS * RS_UNROLL / 2) s *= 2;
    }
    if (s > RS_MAX_SPLITS) s = RS_MAX_SPLITS;
    if (s > chunks) s = chunks;
    return s < 1 ? 1 : s;
}
