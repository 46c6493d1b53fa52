// Ceiling probe: what the matrix pipes sustain on this chip with no memory traffic at all, and with the
// GEMM's LDS read mix beside them.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int RANDOM>
__global__ __launch_bounds__(512, 1) void peak_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 a[6], b[4];
    unsigned h = (threadIdx.x + blockIdx.x * 512) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((int)(h >> 8) % 2001 - 1000) * 0.001f; };
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(RANDOM ? rnd() : (float)(lane + i + j));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(RANDOM ? rnd() : (float)(lane * 2 + i + j));
    if (RANDOM && MODE == 1)
        for (int i = threadIdx.x; i < 16384; i += 512) ((unsigned*)smem)[i] = (__builtin_bit_cast(unsigned, rnd()) & 0xFFFF0000u) | (__builtin_bit_cast(unsigned, rnd()) >> 16);
    __syncthreads();
    f32x4 acc[6][4];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f32x16 acc32[3][2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0;
    const unsigned char* base = smem + (threadIdx.x * 16) % 65536;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 24 x 16x16x32, registers only
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else if (MODE == 1) {   // 24 x 16x16x32 + 10 ds_read_b128 (the GEMM's ratio)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[i] = *(const bf16x8*)(base + i * 1024 + (it & 7) * 8192);
                if (i < 4) b[i] = *(const bf16x8*)(base + (i + 6) * 1024 + (it & 7) * 8192);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if (MODE == 3) {   // mode 0 in snake order: consecutive MFMAs share one operand register
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = (i & 1) ? 3 - jj : jj;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        } else if (MODE == 4) {   // mode 0 with both operands changing at every step (diagonal walk)
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int j = (i + d) & 3;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        } else {   // 12 x 32x32x16, registers only (same flops as mode 0)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + r], b[j + r], acc32[i][j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) s += acc32[i][j][0] + acc32[i][j][7];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int RANDOM>
void run(const char* name, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((peak_kernel<MODE, RANDOM>), dim3(256 * 4), dim3(512), 65536, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 1024.0 * 8 * iters * 24 * 16384.0;
        printf("%-28s iters %d  %.3f ms  %.1f TFLOP/s\n", name, iters, ms, flops / ms * 1e-9);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out;
    hipMalloc(&out, 4);
    run<0, 0>("16x16x32 regs only", iters * 4, out);
    run<0, 1>("16x16x32 regs only, random", iters * 4, out);
    run<3, 1>("16x16x32 random, snake order", iters * 4, out);
    run<4, 1>("16x16x32 random, diagonal", iters * 4, out);
    run<2, 1>("32x32x16 regs only, random", iters * 4, out);
    run<1, 0>("16x16x32 + ds_read_b128", iters * 4, out);
    run<1, 1>("16x16x32 + ds_read_b128, rnd", iters * 4, out);
    return 0;
}
