// Structure probe (not on the product path): a bf16 GEMM main loop with ONE wave per SIMD -- four waves per workgroup as
// 2(M) x 2(N), 192 x 256 x 64 tile, each wave 96 x 128 outputs in 192 accumulator registers, both K slices of a tile's
// fragments held in registers (two sets: the MFMAs of tile t run from one while the ds_reads of tile t + 1 fill the other),
// one barrier per K tile, operands by LDS-DMA into two stages.  It answers one question: does a single wave that
// interleaves its own DMA issue and LDS reads between its MFMAs beat the shipped loop's two waves per SIMD that alternate
// roles between barriers?  C = A[M,K] W[N,K]^T, M % 192 == 0, N % 256 == 0, K % 128 == 0.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gemm4w_probe.hip -o tools/gemm4w_probe ; run: ./tools/gemm4w_probe [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#ifndef PROBE_ABLATE
#define PROBE_ABLATE 0      // timing only (results garbage): 1 no DMA in the loop, 2 no fragment reads in the loop, 4 no barrier, 8 no wait for the DMA, 16 every other DMA piece issued in the first half (races with the reads)
#endif
#ifndef PROBE_BUFFER
#define PROBE_BUFFER 0      // 1: the DMA as buffer_load ... lds (descriptor + 32-bit lane offset + scalar K offset)
#endif
constexpr int BM = 192, BN = 256, BK = 64, ROWB = 128;
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;   // 24 + 32 = 56 KiB

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_uniform, 16, 0, 0);
}

__device__ __forceinline__ void tile_of_block(int bid, int nt, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GM = 8;
    const int band = wg / (GM * tiles_n);
    const int first_m = band * GM;
    const int rows = min(GM, tiles_m - first_m);
    const int in_band = wg - band * GM * tiles_n;
    tm = first_m + in_band % rows;
    tn = in_band / rows;
}

template <int STORE>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1)))
gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = K / BK;

    // DMA sources: piece p (8 rows x 128 B): A pieces 0..23, B pieces 0..31; wave w moves pieces w, w + 4, ...
    const bf16_t* a_src[6];
    const bf16_t* b_src[8];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[i] = A + (size_t)(m0 + row) * K + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        b_src[i] = W + (size_t)(n0 + row) * K + chunk * 8;
    }
    const int fr = lane & 15, fq = lane >> 4, swz = fr >> 1;
    const int a_off = (wm * 96 + fr) * ROWB;
    const int b_off = A_BYTES + (wn * 128 + fr) * ROWB;
    const int c00 = ((0 + fq) ^ swz) * 16, c01 = ((4 + fq) ^ swz) * 16;

    f32x4 acc[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 FA[2][6], FB[2][8];    // [K slice][fragment]: slice 0 / 1 of the tile being multiplied, refilled half a tile ahead

#if PROBE_BUFFER
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    unsigned a_voff[6], b_voff[8];
#pragma unroll
    for (int i = 0; i < 6; ++i) a_voff[i] = (unsigned)((const unsigned char*)a_src[i] - (const unsigned char*)A);
#pragma unroll
    for (int i = 0; i < 8; ++i) b_voff[i] = (unsigned)((const unsigned char*)b_src[i] - (const unsigned char*)W);
    auto issue_piece_a = [&](int i, int kt, int st) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(smem + st * STAGE + (wave + 4 * i) * 1024), 16, a_voff[i], kt * (BK * 2), 0, 0);
    };
    auto issue_piece_b = [&](int i, int kt, int st) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(smem + st * STAGE + A_BYTES + (wave + 4 * i) * 1024), 16, b_voff[i], kt * (BK * 2), 0, 0);
    };
#else
    auto issue_piece_a = [&](int i, int kt, int st) { glds16(a_src[i] + (size_t)kt * BK, smem + st * STAGE + (wave + 4 * i) * 1024); };
    auto issue_piece_b = [&](int i, int kt, int st) { glds16(b_src[i] + (size_t)kt * BK, smem + st * STAGE + A_BYTES + (wave + 4 * i) * 1024); };
#endif
    auto read_a = [&](int ks, int mf, int st) {
        FA[ks][mf] = *(const bf16x8*)(smem + st * STAGE + a_off + mf * 16 * ROWB + (ks ? c01 : c00));
    };
    auto read_b = [&](int ks, int nf, int st) {
        FB[ks][nf] = *(const bf16x8*)(smem + st * STAGE + b_off + nf * 16 * ROWB + (ks ? c01 : c00));
    };

    // prologue: tiles 0 and 1 in flight, K slice 0 of tile 0 in registers
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_piece_a(i, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece_b(i, 0, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_piece_a(i, 1, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece_b(i, 1, 1);
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mf = 0; mf < 6; ++mf) read_a(0, mf, 0);
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) read_b(0, nf, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // The MFMAs are inline assembly with the accumulators pinned to the accumulation registers ("+a"): left to the
    // compiler, 192 accumulators + 112 fragment registers make it shuffle values between the two register files at every
    // loop boundary (250 moves per two K tiles).  Groups are fenced with sched_barrier so that each ds_read / DMA stays
    // between the MFMAs it is written between.
#define MFMA48(ks, first, count)                                                                             \
    _Pragma("unroll") for (int idx = (first); idx < (first) + (count); ++idx) {                              \
        const int mf = idx / 8, nn = idx % 8;                                                                \
        const int nf = (mf & 1) ? 7 - nn : nn;                                                               \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mf][nf]) : "v"(FB[ks][nf]), "v"(FA[ks][mf])); \
    }
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define KTILE(cur, t)                                                                                        \
    do {                                                                                                     \
        const int kt2 = min((t) + 2, nk - 1);                                                                \
        _Pragma("unroll") for (int g = 0; g < 14; ++g) {                                                     \
            if ((PROBE_ABLATE & 16) && (g & 1)) { if (g < 6) issue_piece_a(g, kt2, (cur)); else issue_piece_b(g - 6, kt2, (cur)); } \
            if (!(PROBE_ABLATE & 2)) { if (g < 6) read_a(1, g, (cur)); else read_b(1, g - 6, (cur)); }       \
            FENCE();                                                                                         \
            MFMA48(0, g * 3, 3);                                                                             \
            FENCE();                                                                                         \
        }                                                                                                    \
        MFMA48(0, 42, 6);                                                                                    \
        if (PROBE_ABLATE & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                     \
        if (!(PROBE_ABLATE & 4)) __builtin_amdgcn_s_barrier();                                               \
        FENCE();                                                                                             \
        _Pragma("unroll") for (int g = 0; g < 14; ++g) {                                                     \
            if (!(PROBE_ABLATE & 1) && !((PROBE_ABLATE & 16) && (g & 1))) { if (g < 6) issue_piece_a(g, kt2, (cur)); else issue_piece_b(g - 6, kt2, (cur)); } \
            FENCE();                                                                                         \
            MFMA48(1, g * 3, 1);                                                                             \
            FENCE();                                                                                         \
            if (!(PROBE_ABLATE & 2)) { if (g < 6) read_a(0, g, (cur) ^ 1); else read_b(0, g - 6, (cur) ^ 1); } \
            FENCE();                                                                                         \
            MFMA48(1, g * 3 + 1, 2);                                                                         \
            FENCE();                                                                                         \
        }                                                                                                    \
        MFMA48(1, 42, 6);                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        FENCE();                                                                                             \
    } while (0)

    for (int t = 0; t < nk; t += 2) {
        KTILE(0, t);
        KTILE(1, t + 1);
    }
#undef KTILE
#undef MFMA48
#undef FENCE
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs' results must be in the register file before they are read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (STORE) {
        // plain (slow) write-out for the correctness check: lane holds C[m][n .. n+3], m = fr, n = 4 fq
#pragma unroll
        for (int mf = 0; mf < 6; ++mf)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) {
                const int m = m0 + wm * 96 + mf * 16 + fr;
                const int n = n0 + wn * 128 + nf * 16 + fq * 4;
                *(f32x4*)(C + (size_t)m * N + n) = acc[mf][nf];
            }
    } else {
        float s = 0.f;
#pragma unroll
        for (int mf = 0; mf < 6; ++mf)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) s += acc[mf][nf][0] + acc[mf][nf][1] + acc[mf][nf][2] + acc[mf][nf][3];
        if (s == 12345.678f) C[0] = s;
    }
}

static bf16_t f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (bf16_t)(u >> 16);
}
static float bf2f(bf16_t b) {
    unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 3072, N = argc > 2 ? atoi(argv[2]) : 24576, K = argc > 3 ? atoi(argv[3]) : 4096;
    if (M % BM || N % BN || K % (2 * BK)) { printf("shape must be a multiple of %d x %d x %d\n", BM, BN, 2 * BK); return 1; }
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    unsigned h = 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((int)(h >> 8) % 2001 - 1000) * 0.001f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd() * 0.05f);
    bf16_t *dA, *dW;
    float* dC;
    hipMalloc(&dA, hA.size() * 2);
    hipMalloc(&dW, hW.size() * 2);
    hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    const int lds = 2 * STAGE;
    hipFuncSetAttribute((const void*)gemm4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)gemm4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = (M / BM) * (N / BN);
    hipLaunchKernelGGL(gemm4w_kernel<1>, dim3(grid), dim3(256), lds, 0, dA, dW, dC, M, N, K);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 200; ++s) {
        h = h * 1664525u + 1013904223u;
        const int m = (h >> 8) % M;
        h = h * 1664525u + 1013904223u;
        const int n = (h >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
        worst = fmax(worst, fabs(ref - hC[(size_t)m * N + n]) / (fabs(ref) + 1e-3));
    }
    printf("check: worst relative error over 200 samples %.2e %s\n", worst, worst < 1e-3 ? "(ok)" : "(WRONG)");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(grid), dim3(256), lds, 0, dA, dW, dC, M, N, K);
        hipEventRecord(e0, 0);
        const int n = 30;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(grid), dim3(256), lds, 0, dA, dW, dC, M, N, K);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("M=%d N=%d K=%d  one wave per SIMD, main loop only: %.1f us per launch, %.0f TFLOP/s\n", M, N, K, ms / n * 1e3,
               2.0 * M * N * K / (ms / n * 1e-3) * 1e-12);
    }
    return 0;
}
