// Bidirectional multi-head attention for gfx950, head_dim 128, bf16 in/out, float32 online softmax.
//
// Replaces torch's scaled_dot_product_attention(is_causal=False) inside the mask predictor the
// reference calls at generate.py:88,92 / get_log_likelihood.py:38: every denoising step attends
// the whole sequence, so there is no causal mask, only an optional key-padding mask
// (generate.py:63-64).
//
// One workgroup = NW waves (4 unless the sequence is shorter than 128), 32 queries per wave, all
// of one (sequence, head).  Per 64-key tile: S^T = K Q^T on v_mfma_f32_32x32x16_bf16 (keys on the
// accumulator rows, so a query's scores sit in one lane's registers and the row max/sum need no
// cross-lane work beyond one half-swap), P is converted to bf16 in registers and used directly as
// the B operand of O^T += V^T P^T; the V^T fragments come out of the row-major V tile in LDS
// through the transposing ds_read_b64_tr_b16.
//
// K/V tiles are brought in by 16-byte LDS-DMA into two buffers, one barrier per tile: the next
// tile's DMA is in flight while the current one is computed and no registers are spent on staging.
// LDS rows are 256 B, unpadded; bank conflicts are avoided by XOR swizzles applied on the DMA
// source address and on the read address:
//   K (ds_read_b128, lanes = 32 keys x 2 chunks):   16-byte chunk c of row k sits at c ^ (k & 15)
//   V (ds_read_b64_tr_b16, lanes = 4 keys x 64 B):  64-byte slot s of row k sits at s ^ (k & 3)
#include <type_traits>

#include "common.h"

namespace {

constexpr int HD = 128;
constexpr int KT = 64;          // keys per tile

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
}

// SPLIT: blockIdx.z = sequence * key_splits + split; the workgroup sweeps only its own run of key tiles and
// leaves an unnormalised partial result (float32 O, running max in the log2 domain, row sum) for
// attn_combine_kernel.  Used when there are too few (query block, head, sequence) triples to fill the chip.
// MODE 0: the whole key range; 1 = SPLIT; 2 = SPLIT with one tile per run (a single K/V buffer, so that four
// one-wave workgroups fit in a CU's LDS).
// Diagnostic build (-DLLADA_ATTN_TRACE=1): workgroup 0 of the unsplit kernel stamps five points of every key tile of every
// wave with the shader clock (LDS) and copies them to the workspace the caller passes (tools/attn_trace.py).
#ifndef LLADA_ATTN_TRACE
#define LLADA_ATTN_TRACE 0
#endif
#if LLADA_ATTN_TRACE
#define ATTN_STAMP(i)                                                                                  \
    do {                                                                                               \
        if (!SPLIT && blockIdx.x == 0 && t < 32 && lane == 0)                                          \
            trace_lds[(wave * 32 + t) * 8 + (i)] = (unsigned)__builtin_readcyclecounter();             \
    } while (0)
#else
#define ATTN_STAMP(i)
#endif

template <int NW, int MODE>
__global__ void __launch_bounds__(NW * 64, (NW >= 4 ? 2 : 1))
attn_bidir_kernel(const bf16_t* __restrict__ q_src, int64_t q_stride, int n_q, const bf16_t* __restrict__ kv_src,
                      int64_t kv_stride, const uint8_t* __restrict__ key_mask, bf16_t* __restrict__ out, int L, int d,
                      float scale_log2e, int key_splits, int tiles_per_split, float* __restrict__ part_o,
                      float* __restrict__ part_ml, int64_t n_rows, const bf16_t* __restrict__ blk_kv, int blk_start,
                      int blk_len, int n_qblocks, int n_heads) {
    constexpr bool SPLIT = MODE != 0;
    constexpr int NBUF = MODE == 2 ? 1 : 2;
    constexpr int TILE_BYTES = KT * 256;           // one operand, one tile
    constexpr int BUF_BYTES = 2 * TILE_BYTES;      // K then V
    __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * BUF_BYTES + NBUF * KT * 4];
    float* bias_lds = (float*)(lds + NBUF * BUF_BYTES);
#if LLADA_ATTN_TRACE
    __shared__ unsigned trace_lds[NW * 32 * 8];
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hg = lane >> 5;
    // Work item of this block.  Consecutive blocks are dealt round-robin to the 8 XCDs, each with an L2 of
    // its own, so the query blocks of one (sequence, head) -- which all stream the same K/V rows -- must be
    // CONSECUTIVE ITEMS OF ONE XCD, or every one of them pulls its own copy of K/V over the fabric (measured:
    // without the K/V traffic the kernel is twice as fast).  XCD x works on a contiguous run of items (bijective
    // for any item count); an item is (query block, head, sequence [x key split]), query block fastest.
    int qb, h, bz;
    {
        const int nt = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, per = nt >> 3, rem = nt & 7;
        const int item = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (bid >> 3);
        qb = item % n_qblocks;
        const int rest = item / n_qblocks;
        h = rest % n_heads;
        bz = rest / n_heads;
    }
    const int b = SPLIT ? bz / key_splits : bz;
    const int ks = SPLIT ? bz % key_splits : 0;
    const int q0 = qb * (NW * 32) + wave * 32;
    // queries: n_q rows per sequence (row stride q_stride, head h at column h * HD); keys and values: L rows per
    // sequence of a [.. | K | V] image (row stride kv_stride, K at column d, V at column 2 d)
    const size_t row_stride = (size_t)kv_stride;
    const bf16_t* base = kv_src + (size_t)b * L * row_stride + (size_t)h * HD;
    // optional override: keys blk_start .. blk_start + blk_len - 1 of every sequence come from the compact rows
    // blk_kv [n_seq * blk_len, kv_stride] (the block-cached step's fresh keys/values) instead of the image
    const bf16_t* blk_base = blk_kv ? blk_kv + (size_t)b * blk_len * row_stride + (size_t)h * HD : nullptr;

    // ---- DMA: a tile is 32 pieces of 1 KiB (4 keys x 256 B): pieces 0-15 = K, 16-31 = V ----
    const int p_row = lane >> 4, p_slot = lane & 15;
    auto issue_piece = [&](int kt0, int buf, int piece) {   // piece is wave-uniform
        const int is_v = piece >> 4;
        const int row = (piece & 15) * 4 + p_row;
        const int chunk = is_v ? ((((p_slot >> 2) ^ (row & 3)) << 2) | (p_slot & 3)) : (p_slot ^ (row & 15));
        const int key = min(kt0 + row, L - 1);
        const bool in_blk = blk_base != nullptr && key >= blk_start && key < blk_start + blk_len;
        const bf16_t* row_ptr = in_blk ? blk_base + (size_t)(key - blk_start) * row_stride : base + (size_t)key * row_stride;
        const bf16_t* src = row_ptr + d + (size_t)is_v * d + chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(lds + buf * BUF_BYTES + piece * 1024), 16, 0, 0);
    };
    // Source of every piece this wave moves, for the tile at key 0 (NW >= 4 only): a whole tile that does not
    // touch the block override is then issued with one 64-bit add per piece; forming each address from
    // scratch (clamp, override test, 64-bit multiply) costs ~100 vector instructions per tile and wave.
    [[maybe_unused]] const bf16_t* src0[NW >= 4 ? 32 / NW : 1];
    if constexpr (NW >= 4) {
#pragma unroll
        for (int i = 0; i < 32 / NW; ++i) {
            const int piece = wave + i * NW;
            const int is_v = piece >> 4;
            const int row = (piece & 15) * 4 + p_row;
            const int chunk = is_v ? ((((p_slot >> 2) ^ (row & 3)) << 2) | (p_slot & 3)) : (p_slot ^ (row & 15));
            src0[i] = base + (size_t)row * row_stride + d + (size_t)is_v * d + chunk * 8;
        }
    }
    auto issue_tile = [&](int kt0, int buf) {
        if constexpr (NW >= 4) {
            const bool plain = kt0 + KT <= L && (blk_base == nullptr || kt0 + KT <= blk_start || kt0 >= blk_start + blk_len);
            if (plain) {
                const size_t adv = (size_t)kt0 * row_stride;       // wave-uniform
#pragma unroll
                for (int i = 0; i < 32 / NW; ++i)
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src0[i] + adv),
                                                     (lds_void_ptr)(lds + buf * BUF_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 32 / NW; ++i) issue_piece(kt0, buf, wave + i * NW);
            }
        } else {   // short sequences only: a rolled loop keeps the 16-32 source addresses out of registers
#pragma unroll 1
            for (int piece = wave; piece < 32; piece += NW) issue_piece(kt0, buf, piece);
        }
    };
    auto write_bias = [&](int kt0, int buf) {
        if (tid < KT) {
            const int key = kt0 + tid;
            bool ok = key < L;
            if (ok && key_mask) ok = key_mask[(size_t)b * L + key] != 0;
            bias_lds[buf * KT + tid] = ok ? 0.f : -INFINITY;
        }
    };

    const int n_tiles_all = (L + KT - 1) / KT;
    const int t_begin = SPLIT ? ks * tiles_per_split : 0;
    const int n_tiles = SPLIT ? min(n_tiles_all, t_begin + tiles_per_split) : n_tiles_all;   // one past the last tile
    issue_tile(t_begin * KT, 0);
    write_bias(t_begin * KT, 0);

    // Q fragments: B operand of S^T = K Q^T, lane holds Q[q0 + r][16*ds + 8*hg .. +7]
    bf16x8 qf[8];
    {
        const int q = min(q0 + r, n_q - 1);
        const bf16_t* qp = q_src + ((size_t)b * n_q + q) * (size_t)q_stride + (size_t)h * HD + hg * 8;
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) qf[ds] = *(const bf16x8*)(qp + ds * 16);
    }

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // lane constants of the swizzled reads
    const int k_row_off = r * 256;
    const int k_xor = (hg ^ (r & 15)) * 16;                       // chunk (2 ds + hg) ^ (r & 15) = (2 ds) ^ this
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int v_row_off = (4 * hg + q4) * 256 + 32 * g1 + 8 * p4;  // + key block; (row & 3) == q4
    const int v_xor = q4 * 64;

    for (int t = t_begin; t < n_tiles; ++t) {
        const int buf = (t - t_begin) & (NBUF - 1);
        ATTN_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t (and Q) have landed
        ATTN_STAMP(1);
        __syncthreads();                                    // ... everyone's; and tile t-1 is no longer read
        ATTN_STAMP(2);
        // The next tile's DMA: a wave stands still for ~50 cycles per 1-KiB piece while its instructions queue up in
        // the CU's memory path (8 pieces: a quarter of the QK phase with the matrix pipe idle), so the four-wave kernel
        // issues one piece behind each pair of QK MFMAs instead of all eight up front.
        [[maybe_unused]] bool dma_in_loop = false;
        [[maybe_unused]] size_t dma_adv = 0;
        if constexpr (NBUF == 2) {
            if (t + 1 < n_tiles) {
                const int kt1 = (t + 1) * KT;
                write_bias(kt1, buf ^ 1);
                if constexpr (NW == 4) {
                    dma_in_loop = kt1 + KT <= L && (blk_base == nullptr || kt1 + KT <= blk_start || kt1 >= blk_start + blk_len);
                    dma_adv = (size_t)kt1 * row_stride;
                    if (!dma_in_loop) issue_tile(kt1, buf ^ 1);
                } else {
                    issue_tile(kt1, buf ^ 1);
                }
            }
        }
        const unsigned char* k_lds = lds + buf * BUF_BYTES;
        const unsigned char* v_lds = k_lds + TILE_BYTES;
        const int kt0 = t * KT;

        // ---- S^T[key][q] for two 32-key blocks ----
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
        // The K fragments are read two steps ahead of the MFMAs that use them (three register sets, order pinned
        // with scheduling groups): left to itself the compiler reuses one set and every pair of MFMAs waits out a
        // full LDS round trip.
        bf16x8 kf[3][2];
        auto read_k = [&](int ds) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                kf[ds % 3][kb] = *(const bf16x8*)(k_lds + kb * 32 * 256 + k_row_off + ((ds * 32) ^ k_xor));
        };
        auto qk = [&](auto with_dma) {
            read_k(0);
            read_k(1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int ds = 0; ds < 8; ++ds) {
                if (ds + 2 < 8) {
                    read_k(ds + 2);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds % 3][kb], qf[ds], s[kb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if constexpr (decltype(with_dma)::value && NW == 4) {
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src0[ds] + dma_adv),
                                                     (lds_void_ptr)(lds + (buf ^ 1) * BUF_BYTES + (wave + ds * NW) * 1024), 16, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        };
        if constexpr (NW == 4) {
            if (dma_in_loop)
                qk(std::true_type{});
            else
                qk(std::false_type{});
        } else {
            qk(std::false_type{});
        }
        ATTN_STAMP(3);
        // The additive key bias is 0 or -inf, so it can be added before the scaling; the scale and the
        // running max are then folded into one FMA per score: p = 2^(s * scale - m).
        const bool need_bias = (key_mask != nullptr) || (kt0 + KT > L);
        if (need_bias) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 bias = *(const f32x4*)(bias_lds + buf * KT + kb * 32 + 8 * c + 4 * hg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[kb][4 * c + e] += bias[e];
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kb][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;
        // Lazy rescaling: the reference maximum of a row moves only when its running maximum has risen by more than
        // 2^8 since the last move (any row of the wave: the branch is wave-uniform).  Until then the probabilities are
        // taken against the older maximum and may reach 2^8, which float32 sums and bf16 operands hold as exactly as
        // values below 1; the 64 multiplies of the output accumulators and the extra exponential are skipped on
        // almost every tile after the first.
        const float m_cand = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64(m_cand > m_run + 8.0f) != 0) {
            const float m_to = (m_cand == -INFINITY) ? 0.f : m_cand;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_to);
            m_run = m_cand;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
        }
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
        const f32x2 scale2 = {scale_log2e, scale_log2e}, negm2 = {-m_use, -m_use};
        f32x2 psum2 = {0.f, 0.f};
        bf16x8 pb[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 t = __builtin_elementwise_fma((f32x2){s[kb][e], s[kb][e + 1]}, scale2, negm2);   // one packed FMA
                const f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                psum2 += pv;
                pb[kb][e >> 3][e & 7] = (__bf16)pv[0];
                pb[kb][e >> 3][(e & 7) + 1] = (__bf16)pv[1];
            }
        }
        l_run += psum2[0] + psum2[1];

        ATTN_STAMP(4);
        // ---- O^T[dd][q] += V^T[dd][key] P^T[key][q] ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const unsigned char* vk = v_lds + (kb * 32 + 16 * ss) * 256 + v_row_off;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const int so = (db * 64) ^ v_xor;
                    const s16x4 lo = tr_read(vk + so);
                    const s16x4 hi = tr_read(vk + so + 8 * 256);
                    const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // register pairs, no ALU
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pb[kb][ss], o[db],
                                                                    0, 0, 0);
                }
            }
        }
        ATTN_STAMP(5);
    }
#if LLADA_ATTN_TRACE
    if (!SPLIT && blockIdx.x == 0 && part_o) {
        __syncthreads();
        for (int i = tid; i < NW * 32 * 8; i += NW * 64) ((unsigned*)part_o)[i] = trace_lds[i];
    }
#endif

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const int q = q0 + r;
    if constexpr (SPLIT) {
        if (q < n_q) {
            const size_t row = (size_t)b * n_q + q;
            float* po = part_o + ((size_t)ks * n_rows + row) * d + (size_t)h * HD + 4 * hg;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *(f32x4*)(po + db * 32 + 8 * c) = (f32x4){o[db][4 * c], o[db][4 * c + 1], o[db][4 * c + 2], o[db][4 * c + 3]};
            if (hg == 0) {
                float* pm = part_ml + (((size_t)ks * n_rows + row) * (d / HD) + h) * 2;
                pm[0] = m_run;
                pm[1] = l_tot;
            }
        }
        return;
    }
    // Write-out.  A lane holds 16 pieces of 4 dimensions of ONE query row: stored from the registers, every wave
    // instruction would put 8 bytes into each of 32 rows at a 2d-byte stride (64 partial cache lines, 16 times).
    // The wave's 32 x 128 result goes through its own slice of the K/V buffers (now idle) instead and leaves as
    // 16 bytes per lane, 16 lanes per row: every instruction stores four whole 256-byte head rows.
    if constexpr (!SPLIT) {
    const float inv = 1.0f / l_tot;
    constexpr int OROW = HD * 2 + 16;                      // LDS row stride: 16-byte aligned, 2-way at worst on the 8-byte writes
    static_assert(4 * 32 * OROW <= NBUF * BUF_BYTES, "the write-out slices must fit the K/V buffers");
    __syncthreads();                                       // every wave has finished reading the last K/V tile
    unsigned char* my = lds + wave * (32 * OROW);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x2 w;
            w[0] = pack_bf16x2(o[db][4 * c] * inv, o[db][4 * c + 1] * inv);
            w[1] = pack_bf16x2(o[db][4 * c + 2] * inv, o[db][4 * c + 3] * inv);
            *(u32x2*)(my + r * OROW + (db * 32 + 8 * c + 4 * hg) * 2) = w;
        }
    }
    // the slice is private to the wave: its own LDS writes are ordered before its own reads
    const int orow = lane >> 4, ochunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = i * 4 + orow;
        const u32x4 v = *(const u32x4*)(my + rl * OROW + ochunk * 16);
        const int qo = q0 + rl;
        if (qo < n_q) *(u32x4*)(out + ((size_t)b * n_q + qo) * d + (size_t)h * HD + ochunk * 8) = v;
    }
    }
}

// Merges the key-split partial results: out = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s, m = max_s m_s.
// One thread per 4 output columns.
__global__ void __launch_bounds__(256) attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           bf16_t* __restrict__ out, int64_t n_rows, int H, int key_splits) {
    const int d = H * HD;
    const int64_t total = n_rows * (d / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (d / 4);
        const int col = (int)(i % (d / 4)) * 4;
        const int h = col / HD;
        float m = -INFINITY;
        for (int s = 0; s < key_splits; ++s) m = fmaxf(m, part_ml[((s * n_rows + row) * H + h) * 2]);
        const float m_use = (m == -INFINITY) ? 0.f : m;
        float l = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < key_splits; ++s) {
            const float* pm = part_ml + ((s * n_rows + row) * H + h) * 2;
            const float w = __builtin_amdgcn_exp2f(pm[0] - m_use);
            const f32x4 o = *(const f32x4*)(part_o + (s * n_rows + row) * d + col);
            l += w * pm[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * o[e];
        }
        const float inv = 1.0f / l;
        u32x2 w2;
        w2[0] = pack_bf16x2(acc[0] * inv, acc[1] * inv);
        w2[1] = pack_bf16x2(acc[2] * inv, acc[3] * inv);
        *(u32x2*)(out + row * d + col) = w2;
    }
}

template <int NW>
void launch_attn(const bf16_t* q, int64_t q_stride, int n_q, const bf16_t* kv, int64_t kv_stride, const uint8_t* key_mask,
                 bf16_t* out, int n_seq, int L, int H, int d, float scale, int key_splits, int tiles_per_split,
                 float* part_o, float* part_ml, const bf16_t* blk_kv, int blk_start, int blk_len, hipStream_t stream) {
    const int64_t n_rows = (int64_t)n_seq * n_q;
    if (key_splits > 1) {
        const int nqb = (n_q + NW * 32 - 1) / (NW * 32);
        dim3 grid(nqb * H * n_seq * key_splits);
        if (tiles_per_split == 1)
            hipLaunchKernelGGL((attn_bidir_kernel<NW, 2>), grid, dim3(NW * 64), 0, stream, q, q_stride, n_q, kv, kv_stride,
                               key_mask, out, L, d, scale, key_splits, tiles_per_split, part_o, part_ml, n_rows, blk_kv,
                               blk_start, blk_len, nqb, H);
        else
            hipLaunchKernelGGL((attn_bidir_kernel<NW, 1>), grid, dim3(NW * 64), 0, stream, q, q_stride, n_q, kv, kv_stride,
                               key_mask, out, L, d, scale, key_splits, tiles_per_split, part_o, part_ml, n_rows, blk_kv,
                               blk_start, blk_len, nqb, H);
        const int64_t items = n_rows * (d / 4);
        const int cgrid = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
        hipLaunchKernelGGL(attn_combine_kernel, dim3(cgrid), dim3(256), 0, stream, (const float*)part_o,
                           (const float*)part_ml, out, n_rows, H, key_splits);
        return;
    }
    const int nqb = (n_q + NW * 32 - 1) / (NW * 32);
    dim3 grid(nqb * H * n_seq);
    hipLaunchKernelGGL((attn_bidir_kernel<NW, 0>), grid, dim3(NW * 64), 0, stream, q, q_stride, n_q, kv, kv_stride,
                       key_mask, out, L, d, scale, 1, 0, LLADA_ATTN_TRACE ? part_o : (float*)nullptr, (float*)nullptr, n_rows, blk_kv, blk_start, blk_len,
                       nqb, H);
}

int g_attn_variant = 0;   // test hook: waves per workgroup (1, 2, 4), 0 = automatic
int g_attn_key_splits = 0;  // test hook: 0 = automatic, 1 = never split, n = split the keys n ways when scratch is given

// Key splits for a launch that would put `wgs` workgroups on the 256 CUs: about 512 workgroups in all
// (measured at 32 queries x 384 keys: 23 -> 12 us for one sequence, 24 -> 19 us for eight; beyond that the
// merge pass and the extra partial results cost more than the parallelism gains).
int pick_key_splits(int64_t wgs, int n_tiles) {
    const int cap = n_tiles < 16 ? n_tiles : 16;
    if (g_attn_key_splits > 0) return g_attn_key_splits < cap ? g_attn_key_splits : cap;
    if (wgs > 256 || n_tiles < 2) return 1;
    const int64_t want = 512 / wgs;
    return (int)(want < cap ? want : cap);
}

size_t attn_scratch_bytes(int64_t n_rows, int H, int key_splits) {
    return (size_t)key_splits * n_rows * ((size_t)H * HD * 4 + (size_t)H * 8);
}

}  // namespace

extern "C" void llada_attn_force_variant(int v) { g_attn_variant = v; }
extern "C" void llada_attn_force_key_splits(int n) { g_attn_key_splits = n; }

static int attn_dispatch(const bf16_t* qp, int64_t q_stride, int n_q, const bf16_t* kvp, int64_t kv_stride,
                         const uint8_t* key_mask, bf16_t* op, int n_seq, int seq_len, int n_heads, void* scratch,
                         size_t scratch_bytes, const bf16_t* blk_kv, int blk_start, int blk_len, hipStream_t st) {
    const int d = n_heads * HD;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    // Waves per workgroup: 4 (128 queries), fewer when a sequence has fewer queries than that.  More waves per
    // workgroup (up to one workgroup per (sequence, head), K/V fetched once) were measured slower at
    // every benchmark shape: the per-tile barrier phase-locks the waves that share a SIMD.
    const int q_waves = (n_q + 31) / 32;
    const int nw = g_attn_variant > 0 ? g_attn_variant : (q_waves >= 4 ? 4 : (q_waves >= 2 ? 2 : 1));
    // Few queries (the block-cached step): cut the keys into runs of tiles, one workgroup each, and merge.
    int key_splits = 1, tiles_per = 0;
    float *part_o = nullptr, *part_ml = nullptr;
    if (scratch) {
        const int n_tiles = (seq_len + KT - 1) / KT;
        const int64_t wgs = (int64_t)((n_q + nw * 32 - 1) / (nw * 32)) * n_heads * n_seq;
        int want = pick_key_splits(wgs, n_tiles);
        const int64_t n_rows = (int64_t)n_seq * n_q;
        while (want > 1 && attn_scratch_bytes(n_rows, n_heads, want) > scratch_bytes) --want;
        if (want > 1 && (int64_t)n_seq * want <= 65535) {
            tiles_per = (n_tiles + want - 1) / want;
            key_splits = (n_tiles + tiles_per - 1) / tiles_per;   // no empty run
            part_o = (float*)scratch;
            part_ml = part_o + (size_t)key_splits * n_rows * d;
        }
    }
    if (LLADA_ATTN_TRACE && scratch && key_splits == 1) part_o = (float*)scratch;   // diagnostic build: the stamps go there
    LLADA_CHECK_SUPPORTED((int64_t)((n_q + nw * 32 - 1) / (nw * 32)) * n_heads * n_seq * key_splits < 0x7fffffffLL,
                          "attn: grid too large");
    switch (nw) {
        case 4: launch_attn<4>(qp, q_stride, n_q, kvp, kv_stride, key_mask, op, n_seq, seq_len, n_heads, d, scale_log2e, key_splits, tiles_per, part_o, part_ml, blk_kv, blk_start, blk_len, st); break;
        case 2: launch_attn<2>(qp, q_stride, n_q, kvp, kv_stride, key_mask, op, n_seq, seq_len, n_heads, d, scale_log2e, key_splits, tiles_per, part_o, part_ml, blk_kv, blk_start, blk_len, st); break;
        default: launch_attn<1>(qp, q_stride, n_q, kvp, kv_stride, key_mask, op, n_seq, seq_len, n_heads, d, scale_log2e, key_splits, tiles_per, part_o, part_ml, blk_kv, blk_start, blk_len, st); break;
    }
    LLADA_CHECK_LAUNCH("attn_bidir_kernel");
    return LLADA_OK;
}

extern "C" int llada_attn_bidir_bf16(const void* qkv, const uint8_t* key_mask, void* out, int32_t n_seq,
                                     int32_t seq_len, int32_t n_heads, void* stream) {
    LLADA_CHECK_ARG(qkv && out, "attn: null operand");
    LLADA_CHECK_ARG(n_seq > 0 && seq_len > 0 && n_heads > 0, "attn: bad shape");
    LLADA_CHECK_SUPPORTED(n_seq <= 65535 && n_heads <= 65535, "attn: grid too large");
    const int64_t stride = (int64_t)3 * n_heads * HD;
    return attn_dispatch((const bf16_t*)qkv, stride, seq_len, (const bf16_t*)qkv, stride, key_mask, (bf16_t*)out, n_seq,
                         seq_len, n_heads, nullptr, 0, nullptr, 0, 0, (hipStream_t)stream);
}

// Attention of a subset of the positions: n_q query rows per sequence (q [n_seq * n_q, q_stride], head h at column
// h * 128) over all seq_len keys and values of a [Q | K | V] image kv [n_seq * seq_len, 3 d] (the per-layer cache of
// the block-cached sampler).  out [n_seq * n_q, d].  block_kv (may be NULL): compact [Q | K | V] rows
// [n_seq * n_q, 3 d] whose K and V stand in for the image's rows block_start .. block_start + n_q - 1 of every
// sequence (the image is not touched).  With a workspace (llada_attn_workspace_bytes; may be NULL) the
// keys are split over several workgroups when the queries alone would leave most CUs idle.
extern "C" size_t llada_attn_workspace_bytes(int32_t n_seq, int32_t n_q, int32_t seq_len, int32_t n_heads) {
    if (n_seq <= 0 || n_q <= 0 || seq_len <= 0 || n_heads <= 0) return 0;
    const int n_tiles = (seq_len + KT - 1) / KT;
    const int splits = n_tiles < 16 ? n_tiles : 16;
    return splits > 1 ? attn_scratch_bytes((int64_t)n_seq * n_q, n_heads, splits) : 0;
}

extern "C" int llada_attn_bidir_queries_bf16(const void* q, int64_t q_stride, int32_t n_q, const void* kv,
                                             const uint8_t* key_mask, void* out, int32_t n_seq, int32_t seq_len,
                                             int32_t n_heads, const void* block_kv, int32_t block_start,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    LLADA_CHECK_ARG(q && kv && out, "attn_queries: null operand");
    LLADA_CHECK_ARG(n_seq > 0 && seq_len > 0 && n_heads > 0 && n_q > 0, "attn_queries: bad shape");
    LLADA_CHECK_ARG(q_stride >= (int64_t)n_heads * HD && (q_stride % 8) == 0 && ((uintptr_t)q & 15) == 0,
                    "attn_queries: q_stride must cover the heads and keep rows 16-byte aligned");
    LLADA_CHECK_SUPPORTED(n_seq <= 65535 && n_heads <= 65535, "attn: grid too large");
    LLADA_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "attn_queries: workspace must be 16-byte aligned");
    LLADA_CHECK_ARG(block_kv == nullptr || (((uintptr_t)block_kv & 15) == 0 && block_start >= 0 && block_start + n_q <= seq_len),
                    "attn_queries: block_kv must be 16-byte aligned and the block must lie inside the sequence");
    return attn_dispatch((const bf16_t*)q, q_stride, n_q, (const bf16_t*)kv, (int64_t)3 * n_heads * HD, key_mask,
                         (bf16_t*)out, n_seq, seq_len, n_heads, workspace, workspace ? workspace_bytes : 0,
                         (const bf16_t*)block_kv, block_start, n_q, (hipStream_t)stream);
}
