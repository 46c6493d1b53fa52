// Philox4x32-10 counter RNG: the library's own random source for the Gumbel noise
// (generate.py:17), the 'random' remasking scores (generate.py:121) and the likelihood masks
// (get_log_likelihood.py:11,20) when the caller injects none.  Keyed by (seed, stream) and
// counted by (index, position, global row, step), so results do not depend on how rows are
// sharded over GPUs.
#pragma once
#include <stdint.h>

#define LLADA_STREAM_GUMBEL 1u
#define LLADA_STREAM_REMASK 2u
#define LLADA_STREAM_LL 3u

struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.v[0] = c0;
    r.v[1] = c1;
    r.v[2] = c2;
    r.v[3] = c3;
    return r;
}

__device__ __forceinline__ void philox_key(uint64_t seed, uint32_t stream, uint32_t& k0, uint32_t& k1) {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32) ^ stream;
}

// 53-bit uniform in [0, 1) from two 32-bit words.
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
