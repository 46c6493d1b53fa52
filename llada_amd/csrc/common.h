// Shared device/host helpers for the gfx950 kernels of libllada_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/llada_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bit pattern

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// Round-to-nearest-even, NaN stays NaN (the compiler emits v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// SiLU in float32: x * 1/(1 + e^-x) with the hardware exponential and reciprocal (1 ulp each); every
// caller rounds the result to bf16 straight away.
__device__ __forceinline__ float silu_f32(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Per-device "done once" flag for function attributes (hipFuncSetAttribute applies to the current
// device only): true the first time it is asked on a device.
struct OncePerDevice {
    bool done[64] = {};
    bool first() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) return true;
        const bool f = !done[dev];
        done[dev] = true;
        return f;
    }
};

// ---- host side error plumbing -------------------------------------------------------------
void llada_set_error(const char* fmt, ...);

#define LLADA_CHECK_ARG(cond, ...)           \
    do {                                     \
        if (!(cond)) {                       \
            llada_set_error(__VA_ARGS__);    \
            return LLADA_ERR_INVALID;        \
        }                                    \
    } while (0)

#define LLADA_CHECK_SUPPORTED(cond, ...)     \
    do {                                     \
        if (!(cond)) {                       \
            llada_set_error(__VA_ARGS__);    \
            return LLADA_ERR_UNSUPPORTED;    \
        }                                    \
    } while (0)

#define LLADA_CHECK_LAUNCH(what)                                                    \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            llada_set_error("%s: %s", what, hipGetErrorString(e_));                 \
            return LLADA_ERR_HIP;                                                   \
        }                                                                           \
    } while (0)

// ---- internal entry points (C++ linkage, not exported) --------------------------------------------
// llada_gemm_bf16 with the rotary table row of every output row given explicitly (positions [M] int32, or
// nullptr for token index % seq_len; values are clamped into [0, seq_len)): the ROPE_QKV epilogue of a
// left-padded batch whose rows count positions from their first real token.
int llada_gemm_bf16_positions(const void* A, const void* W, void* C, int64_t M, int32_t N, int32_t K, int32_t lda,
                              int32_t ldc, int32_t epilogue, const void* aux, int32_t ldr, const float* rope_cos,
                              const float* rope_sin, int32_t seq_len, const int32_t* positions, void* workspace,
                              size_t workspace_bytes, void* stream);
// llada_rope_bf16 with explicit positions (the unfused weight layout).
int llada_rope_bf16_positions(void* qkv, const float* rope_cos, const float* rope_sin, int32_t n_seq, int32_t seq_len,
                              int32_t n_heads, const int32_t* positions, void* stream);
