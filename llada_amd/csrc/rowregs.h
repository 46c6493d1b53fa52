// A full-vocabulary logits row held on chip by one 1024-thread workgroup.
//
// The per-row reductions of the sampler epilogue and of the masked cross-entropy (arg-max,
// soft-max maximum, sum of exponentials, entropy) need two to three sweeps over the row.  One row of
// LLaDA's vocabulary is 126 464 bf16 = 247 KiB: more than the 160 KiB of LDS, so a lane keeps its
// first 8 chunks of 16 bytes in registers (32 VGPRs, 128 KiB per workgroup) and its last 8 in the
// LDS (128 KiB, lane-contiguous 16-byte slots: conflict-free ds_read_b128).  Every lane issues its
// loads up front, memory is read exactly once, and the sweeps then run on chip.  Chunk c of the row
// (8 logits) belongs to lane (c % 1024), slot (c / 1024); slots past the end of the row hold -inf.
#pragma once
#include "common.h"

constexpr int ROW_THREADS = 1024;
constexpr int ROW_SLOTS = 16;                                   // chunks per lane
constexpr int ROW_REG_SLOTS = 8;                                // of which in registers
constexpr int ROW_LDS_BYTES = (ROW_SLOTS - ROW_REG_SLOTS) * ROW_THREADS * 16;   // 131072
constexpr int ROW_MAX_VOCAB = ROW_THREADS * ROW_SLOTS * 8;      // 131072
constexpr unsigned BF16_NEG_INF2 = 0xFF80FF80u;

__device__ __forceinline__ float bf16lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// generate.py:89-90 / get_log_likelihood.py:41-42 on one element, each op rounding to bf16 as torch
// does on bf16 tensors
__device__ __forceinline__ float cfg_combine(float c, float u, float cfg1) {
    float d = round_bf16(c - u);
    d = round_bf16(d * cfg1);
    return round_bf16(u + d);
}

struct RowRegs {
    u32x4 q[ROW_REG_SLOTS];
    u32x4* mine;   // this lane's LDS slots: mine[j * ROW_THREADS], j = 0 .. ROW_SLOTS - ROW_REG_SLOTS - 1

    __device__ __forceinline__ u32x4 get(int i) const { return i < ROW_REG_SLOTS ? q[i] : mine[(i - ROW_REG_SLOTS) * ROW_THREADS]; }
    __device__ __forceinline__ void put(int i, u32x4 v) {
        if (i < ROW_REG_SLOTS)
            q[i] = v;
        else
            mine[(i - ROW_REG_SLOTS) * ROW_THREADS] = v;
    }

    // Loads the row (combined with the unconditional row under classifier-free guidance), in groups of
    // 8 (8 pairs with guidance) 16-byte loads per lane issued back to back.  `lds` is the workgroup's
    // ROW_LDS_BYTES region; only the owning lane ever touches a slot, so no barrier is needed.
    __device__ __forceinline__ void load(const bf16_t* lc, const bf16_t* lu, float cfg1, int chunks, int tid,
                                         unsigned char* lds) {
        mine = (u32x4*)lds + tid;
        const u32x4 ninf = {BF16_NEG_INF2, BF16_NEG_INF2, BF16_NEG_INF2, BF16_NEG_INF2};
#pragma unroll
        for (int g = ROW_SLOTS - 8; g >= 0; g -= 8) {     // the LDS half first, the register half stays
            u32x4 t[8];
            int c[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // unconditional loads (address clamped into the row) travel together
                c[j] = min(tid + (g + j) * ROW_THREADS, chunks - 1);
                t[j] = __builtin_nontemporal_load((const u32x4*)(lc + (size_t)c[j] * 8));
            }
            if (lu) {
                u32x4 b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = __builtin_nontemporal_load((const u32x4*)(lu + (size_t)c[j] * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        t[j][e] = pack_bf16x2(cfg_combine(bf16lo(t[j][e]), bf16lo(b[j][e]), cfg1),
                                              cfg_combine(bf16hi(t[j][e]), bf16hi(b[j][e]), cfg1));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) put(g + j, tid + (g + j) * ROW_THREADS < chunks ? t[j] : ninf);
        }
    }

    // Overwrites vocabulary entry v with -inf in the lane that holds it.
    __device__ __forceinline__ void exclude(int v, int tid) {
        if (v < 0) return;
        const int c = v >> 3, e = v & 7;
        if ((c & (ROW_THREADS - 1)) != tid) return;
        const int slot = c / ROW_THREADS;
#pragma unroll
        for (int i = 0; i < ROW_SLOTS; ++i) {
            if (i != slot) continue;
            u32x4 t = get(i);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w != (e >> 1)) continue;
                t[w] = (e & 1) ? ((t[w] & 0x0000FFFFu) | 0xFF800000u) : ((t[w] & 0xFFFF0000u) | 0x0000FF80u);
            }
            put(i, t);
        }
    }

    __device__ __forceinline__ float max() const {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < ROW_SLOTS; ++i) {
            const u32x4 t = get(i);
#pragma unroll
            for (int w = 0; w < 4; ++w) m = fmaxf(m, fmaxf(bf16lo(t[w]), bf16hi(t[w])));
        }
        return m;
    }

    // Hardware exponential (v_exp_f32, 1 ulp, after the scaling by log2 e): the library expf costs
    // ~10x the instructions and made this sweep, not the memory read, the longest part of the kernel.
    __device__ __forceinline__ float sum_exp(float m) const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ROW_SLOTS; ++i) {
            const u32x4 t = get(i);
#pragma unroll
            for (int w = 0; w < 4; ++w) s += __expf(bf16lo(t[w]) - m) + __expf(bf16hi(t[w]) - m);
        }
        return s;
    }
};

// Workgroup-wide reductions over the 16 waves; `red` is a 16-entry LDS array that is not reused
// before the next __syncthreads() of the caller.
__device__ __forceinline__ float block_max16(float v, float* red, int lane, int wave) {
    v = wave_reduce_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < ROW_THREADS / 64; ++w) m = fmaxf(m, red[w]);
    return m;
}

__device__ __forceinline__ float block_sum16(float v, float* red, int lane, int wave) {
    v = wave_reduce_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ROW_THREADS / 64; ++w) s += red[w];   // fixed order: deterministic
    return s;
}
