// Probe: how fast can the CUs of an MI355X pull operand bytes into LDS by 16-byte LDS-DMA, as a function of
// how many workgroups pull at once and of where the bytes come from?  Decides what a small-M GEMM can hope for:
// its activation tile is re-read by every workgroup (served by the L2 / Infinity Cache), its weights are read
// once (HBM).  No arithmetic, nothing is read back from LDS.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_fill_probe tools/lds_fill_probe.hip && ./lds_fill_probe
//
// mode 0: every workgroup streams the SAME 4 MiB region over and over (an activation matrix, cache-resident)
// mode 1: every workgroup streams its OWN slice of a 2 GiB buffer (weights, HBM)
// mode 2: alternate pieces of both (a GEMM K step: some shared rows, some private rows)
// mode 3: waves 0-3 stream the shared region, waves 4-7 their own slice (the two streams never share a wave)
// mode 4: waves 0-4 shared, waves 5-7 own (5 : 3, about the mix of a 192 x 96 tile)
// Each of the 8 waves keeps `depth` KiB pieces in flight (counted vmcnt).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DEPTH, int AUX = 0>
__global__ void __launch_bounds__(512) fill_kernel(const unsigned char* __restrict__ shared_src, size_t shared_bytes,
                                                   const unsigned char* __restrict__ own_src, size_t own_bytes, int pieces,
                                                   int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* own = own_src + (size_t)blockIdx.x * own_bytes;
    // wave w moves pieces w, w + 8, ...; piece p of a stream = 1 KiB at offset p KiB (wrapping)
    for (int i = 0; i < pieces; ++i) {
        const size_t p = (size_t)i * 8 + wave;
        const bool use_shared = mode == 0 || (mode == 2 && (i & 1)) || (mode == 3 && wave < 4) || (mode == 4 && wave < 5);
        const unsigned char* src = use_shared ? shared_src + ((p * 1024) & (shared_bytes - 1)) : own + ((p * 1024) & (own_bytes - 1));
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane * 16), (lds_ptr_t)(lds + ((i % DEPTH) * 8 + wave) * 1024), 16, 0, AUX);
        if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Waves per workgroup: the shared stream by LDS-DMA from 4, 8 or 16 waves of ONE workgroup per CU (4 KiB in flight each).
template <int NWAVES>
__global__ void __launch_bounds__(NWAVES * 64) waves_kernel(const unsigned char* __restrict__ src, size_t bytes, int pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = 0; i < pieces; ++i) {
        const size_t p = (size_t)i * NWAVES + wave;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ((p * 1024) & (bytes - 1)) + lane * 16),
                                         (lds_ptr_t)(lds + ((i % 4) * NWAVES + wave) * 1024), 16, 0, 0);
        if (i >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The same streams pulled into REGISTERS by plain 16-byte loads (8 in flight per lane), no LDS: tells whether the
// per-CU ceiling above belongs to the LDS-DMA path or to the load path as a whole.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void __launch_bounds__(512) reg_kernel(const unsigned char* __restrict__ shared_src, size_t shared_bytes,
                                                  const unsigned char* __restrict__ own_src, size_t own_bytes, int pieces,
                                                  int mode, unsigned int* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* own = own_src + (size_t)blockIdx.x * own_bytes;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces; i += 8) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t p = (size_t)(i + j) * 8 + wave;
            const bool use_shared = mode == 0 || (mode == 2 && ((i + j) & 1));
            const unsigned char* src = use_shared ? shared_src + ((p * 1024) & (shared_bytes - 1)) : own + ((p * 1024) & (own_bytes - 1));
            v[j] = __builtin_nontemporal_load((const u32x4*)(src + lane * 16));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];   // keeps the loads alive
}

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int DEPTH, int AUX = 0>
float run(int grid, int mode, const unsigned char* shared_src, size_t shared_bytes, const unsigned char* own_src,
          size_t own_bytes, int pieces) {
    const size_t lds_bytes = (size_t)DEPTH * 8 * 1024;
    CHECK(hipFuncSetAttribute((const void*)(fill_kernel<DEPTH, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fill_kernel<DEPTH, AUX>), dim3(grid), dim3(512), lds_bytes, 0, shared_src, shared_bytes, own_src, own_bytes,
                       pieces, mode);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((fill_kernel<DEPTH, AUX>), dim3(grid), dim3(512), lds_bytes, 0, shared_src, shared_bytes, own_src,
                           own_bytes, pieces, mode);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const size_t shared_bytes = 4u << 20;            // about one 384 x 4096 bf16 activation matrix; a power of two (masks, not modulo)
    const size_t own_total = 2ull << 30;             // far beyond the 256 MiB Infinity Cache
    unsigned char *shared_src, *own_src;
    CHECK(hipMalloc(&shared_src, shared_bytes));
    CHECK(hipMalloc(&own_src, own_total));
    CHECK(hipMemset(shared_src, 1, shared_bytes));
    CHECK(hipMemset(own_src, 2, own_total));
    const int pieces = 512;                          // per wave: 512 KiB, 4 MiB per workgroup
    const char* names[5] = {"shared (cache-resident)", "own slice (HBM)", "alternating", "split by wave 4:4", "split by wave 5:3"};
    for (int mode = 0; mode < 5; ++mode)
        for (int grid : {64, 128, 192, 256, 512}) {
            const size_t own_bytes = own_total / 512;   // 4 MiB slices
            const double bytes = (double)grid * 8 * pieces * 1024;
            const float t4 = run<4>(grid, mode, shared_src, shared_bytes, own_src, own_bytes, pieces);
            const float t8 = run<8>(grid, mode, shared_src, shared_bytes, own_src, own_bytes, pieces);
            const float t16 = run<16>(grid, mode, shared_src, shared_bytes, own_src, own_bytes, pieces);
            printf("%-24s grid %3d: depth 4/8/16 KiB per wave: %6.2f / %6.2f / %6.2f TB/s  (%5.1f / %5.1f / %5.1f GB/s per workgroup)\n",
                   names[mode], grid, bytes / t4 / 1e9, bytes / t8 / 1e9, bytes / t16 / 1e9, bytes / t4 / 1e6 / grid,
                   bytes / t8 / 1e6 / grid, bytes / t16 / 1e6 / grid);
            fflush(stdout);
        }
    for (int grid : {128, 256, 512}) {   // the HBM stream with the non-temporal policy (aux = 2) on the DMA
        const size_t own_bytes = own_total / 512;
        const double bytes = (double)grid * 8 * pieces * 1024;
        const float t8 = run<8, 2>(grid, 1, shared_src, shared_bytes, own_src, own_bytes, pieces);
        const float t16 = run<16, 2>(grid, 1, shared_src, shared_bytes, own_src, own_bytes, pieces);
        printf("own slice (HBM), nt DMA    grid %3d: depth 8/16 KiB per wave: %6.2f / %6.2f TB/s\n", grid, bytes / t8 / 1e9, bytes / t16 / 1e9);
    }
    {
        auto time_waves = [&](auto kernel, int nw) {
            const size_t lds_bytes = (size_t)4 * nw * 1024;
            const int pcs = 4096 / nw * 8 / 8;           // 4 MiB per workgroup whatever the wave count
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(kernel, dim3(256), dim3(nw * 64), lds_bytes, 0, shared_src, shared_bytes, pcs);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(256), dim3(nw * 64), lds_bytes, 0, shared_src, shared_bytes, pcs);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            const double bytes = 256.0 * nw * pcs * 1024;
            printf("waves per workgroup %2d (256 workgroups, shared stream): %6.2f TB/s (%5.1f GB/s per CU)\n", nw, bytes / ms / 1e9,
                   bytes / ms / 1e6 / 256);
        };
        time_waves(waves_kernel<2>, 2);
        time_waves(waves_kernel<4>, 4);
        time_waves(waves_kernel<8>, 8);
        time_waves(waves_kernel<16>, 16);
    }
    unsigned int* sink;
    CHECK(hipMalloc(&sink, 4096));
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {64, 128, 256, 512}) {
            const size_t own_bytes = own_total / 512;
            const double bytes = (double)grid * 8 * pieces * 1024;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(reg_kernel, dim3(grid), dim3(512), 0, 0, shared_src, shared_bytes, own_src, own_bytes, pieces, mode, sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r)
                hipLaunchKernelGGL(reg_kernel, dim3(grid), dim3(512), 0, 0, shared_src, shared_bytes, own_src, own_bytes, pieces, mode, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            printf("to registers: %-24s grid %3d: %6.2f TB/s (%5.1f GB/s per workgroup)\n", names[mode], grid, bytes / ms / 1e9,
                   bytes / ms / 1e6 / grid);
            fflush(stdout);
        }
    return 0;
}
