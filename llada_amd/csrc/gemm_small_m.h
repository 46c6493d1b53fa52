// GEMM loop for one or two sequences (M of a few hundred rows), included by gemm.hip inside its namespace.
//
// At M = 384 the 256-column tiles of the main kernels leave a quarter of the CUs idle.  This loop uses a
// 192 x (32 NG)-column tile, NG in {1, 2, 3, 4, 6}, chosen by the host so that one launch has about one work item
// per CU (QKV at M = 384: 192 x 96 -> 256 items), ONE barrier per K step, and a three- or four-stage LDS ring.
//
// 8 waves = 4 (M) x 2 (K): wave (wm, kh) owns rows wm*48 .. +47 of the tile and ALL its columns, and multiplies
// the kh-th 32-deep half of every K step, so every byte of the stage is read from LDS by the fewest waves
// possible (A rows once per K half, W rows four times).  The two K halves of a row block are added through LDS at
// the end (kh = 0 plus kh = 1, a fixed order), and the kh = 0 wave runs the fused epilogue on its 48 rows.
//
// LDS stage: A rows 0..191, then W rows 0..BN-1, 128 bytes (64 k) each, 16-byte chunk c of row r stored at
// chunk c ^ ((r >> 1) & 7) as in the main kernels.  A stage is (192 + BN) / 8 DMA pieces of 1 KiB (8 rows): the
// kh = 0 waves move the 24 activation pieces (6 each), the kh = 1 waves the BN / 8 weight pieces (BN / 32 each),
// so the waits are counted with compile-time constants per role.
//
// What bounds it (measured with the loop reduced to its DMA, tools/sm_kscan.py on -DLLADA_SM_DEBUG builds): a CU's
// vector memory path returns in order, so the activation rows (L2 hits, ~105 GB/s per CU) and the weight rows
// (HBM, ~44 GB/s per CU with every CU streaming) cost their SUM, 0.23 + 0.27 us per K step for a 192 x 96 tile,
// whatever the ring depth and whichever waves issue them; the full loop runs 5-10 % above that.  That is the same
// floor the main kernels sit on at this M (QKV alone: 48 us here, 50 us there; inside a whole denoising step this
// loop is 1-2 % slower), so the launch planner does not select it: it is reachable through
// llada_gemm_force_small_m() only, kept with its exactness test as the measured alternative.
#pragma once

#ifndef LLADA_SM_W_NT
#define LLADA_SM_W_NT 1
#endif
// Diagnostic build (-DLLADA_SM_TRACE=1): workgroup 0 stamps four points of every K step of every wave with the shader
// clock (into 8 KiB of LDS behind the stage ring) and copies them to the split-K workspace when it is done.
// Diagnostic builds (-DLLADA_SM_DEBUG=bits, results are garbage): 1 no DMA of A, 2 no DMA of W, 4 no LDS reads / MFMAs.
#ifndef LLADA_SM_DEBUG
#define LLADA_SM_DEBUG 0
#endif
#ifndef LLADA_SM_TRACE
#define LLADA_SM_TRACE 0
#endif
#if LLADA_SM_TRACE
#define SM_STAMP(i)                                                                                     \
    do {                                                                                                \
        if (blockIdx.x == 0 && t < 64 && lane == 0)                                                     \
            ((unsigned*)(smem + T::LDS_BYTES))[(wave * 64 + t) * 4 + (i)] = (unsigned)__builtin_readcyclecounter(); \
    } while (0)
#else
#define SM_STAMP(i)
#endif
constexpr int SM_BM = 192;
constexpr int SM_A_BYTES = SM_BM * ROW_BYTES;

template <int NG>
struct SmallM {
    static constexpr int BNS = 32 * NG;                    // tile columns
    static constexpr int NF = 2 * NG;                      // 16-column fragments
    static constexpr int B_PIECES = BNS / 8;               // 1-KiB DMA pieces of the weight rows of a stage
    static constexpr int STAGE = (SM_BM + BNS) * ROW_BYTES;
#ifdef LLADA_SM_D
    static constexpr int D = (LLADA_SM_D * STAGE <= 160 * 1024) ? LLADA_SM_D : 3;
#else
    static constexpr int D = (4 * STAGE <= 152 * 1024) ? 4 : 3;
#endif
    static constexpr int LDS_BYTES = D * STAGE;
    static_assert(B_PIECES % 4 == 0, "tile columns must be a multiple of 32");
    static_assert(4 * 3 * NF * 1024 <= LDS_BYTES, "the K-half exchange must fit in the stage ring");
};

// Epilogue of one wave's 48 rows x BN columns (acc[mf][nf]: lane (fr, fq) holds row mf*16 + fr, columns
// nf*16 + fq*4 .. +3).  Same arithmetic as store_tile; the rows leave through the wave's own LDS slice as whole
// 16-byte pieces, (row, piece) pairs dealt to the lanes in order.
template <int NG, int EPI>
__device__ __forceinline__ void store_rows48(const GemmParams& p, f32x4 (&acc)[3][2 * NG], int mrow0, int n0, unsigned char* my,
                                             int lane, int split) {
    constexpr int NF = 2 * NG, BNS = 32 * NG;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (EPI == EPI_PARTIAL) {
        float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) {
            const int m = mrow0 + mf * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n0 + nf * 16 + fq * 4;
                if (n >= p.N) continue;
                *(f32x4*)(slab + (size_t)m * p.N + n) = acc[mf][nf];
            }
        }
    } else {
        constexpr bool SWI = (EPI == LLADA_EPI_SWIGLU);
        constexpr int OUTC = SWI ? BNS / 2 : BNS;
        constexpr int ROWB = OUTC * 2 + 16;
        constexpr int LPR = OUTC / 8;                      // 16-byte pieces per output row
        constexpr int TOTAL = 48 * LPR;
        constexpr int NI = (TOTAL + 63) / 64;
        const int out_n0 = SWI ? n0 / 2 : n0;
        const int out_cols = SWI ? p.N / 2 : p.N;
        [[maybe_unused]] u32x4 resid[NI];
        if constexpr (EPI == LLADA_EPI_RESIDUAL) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int idx = min(i * 64 + lane, TOTAL - 1);
                const int m = min(mrow0 + idx / LPR, (int)p.M - 1);
                const int n = min(out_n0 + (idx % LPR) * 8, out_cols - 8);
                resid[i] = *(const u32x4*)(p.R + (size_t)m * p.ldr + n);
            }
        }
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) {
            unsigned char* row = my + (mf * 16 + fr) * ROWB + fq * 8;
            if constexpr (EPI == LLADA_EPI_PLAIN || EPI == LLADA_EPI_RESIDUAL) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    u32x2 o;
                    o[0] = pack_bf16x2(acc[mf][nf][0], acc[mf][nf][1]);
                    o[1] = pack_bf16x2(acc[mf][nf][2], acc[mf][nf][3]);
                    *(u32x2*)(row + nf * 32) = o;
                }
            } else if constexpr (EPI == LLADA_EPI_ROPE_QKV) {
                const int m = mrow0 + mf * 16 + fr;
                const int pos = rope_row(p, m);
#pragma unroll
                for (int pr = 0; pr < NG; ++pr) {
                    const f32x4 x1 = acc[mf][2 * pr], x2 = acc[mf][2 * pr + 1];
                    f32x4 o1 = x1, o2 = x2;
                    if (n0 + pr * 32 < p.rope_cols) {      // wave-uniform; a tile may straddle the K | V boundary
                        const int j = (((n0 + pr * 32) & 127) >> 5) * 16 + fq * 4;
                        const f32x4 c = *(const f32x4*)(p.rope_cos + (size_t)pos * 64 + j);
                        const f32x4 sn = *(const f32x4*)(p.rope_sin + (size_t)pos * 64 + j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = round_bf16(x1[e]), b = round_bf16(x2[e]);
                            o1[e] = a * c[e] - b * sn[e];
                            o2[e] = b * c[e] + a * sn[e];
                        }
                    }
                    u32x2 w1, w2;
                    w1[0] = pack_bf16x2(o1[0], o1[1]);
                    w1[1] = pack_bf16x2(o1[2], o1[3]);
                    w2[0] = pack_bf16x2(o2[0], o2[1]);
                    w2[1] = pack_bf16x2(o2[2], o2[3]);
                    *(u32x2*)(row + (2 * pr) * 32) = w1;
                    *(u32x2*)(row + (2 * pr + 1) * 32) = w2;
                }
            } else {  // SWIGLU: fragments (2pr, 2pr+1) = (gate, up) of the same 16 hidden units
#pragma unroll
                for (int pr = 0; pr < NG; ++pr) {
                    const f32x4 g = acc[mf][2 * pr], u = acc[mf][2 * pr + 1];
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gb = round_bf16(g[e]);
                        const float act = round_bf16(silu_f32(gb));
                        o[e] = act * round_bf16(u[e]);
                    }
                    u32x2 w;
                    w[0] = pack_bf16x2(o[0], o[1]);
                    w[1] = pack_bf16x2(o[2], o[3]);
                    *(u32x2*)(row + pr * 32) = w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = i * 64 + lane;
            const int rl = min(idx, TOTAL - 1) / LPR, ch = min(idx, TOTAL - 1) % LPR;
            const int m = mrow0 + rl, n = out_n0 + ch * 8;
            u32x4 v = *(const u32x4*)(my + rl * ROWB + ch * 16);
            if constexpr (EPI == LLADA_EPI_RESIDUAL) {
                const u32x4 r = resid[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf16_to_f32(r[e] & 0xFFFF) + bf16_to_f32(v[e] & 0xFFFF);
                    const float hi = bf16_to_f32(r[e] >> 16) + bf16_to_f32(v[e] >> 16);
                    v[e] = pack_bf16x2(lo, hi);
                }
            }
            if (idx < TOTAL && m < p.M && n < out_cols) {
                u32x4* dst = (u32x4*)(p.C + (size_t)m * p.ldc + n);
                if constexpr (EPI == LLADA_EPI_RESIDUAL)
                    *dst = v;
                else
                    __builtin_nontemporal_store(v, dst);
            }
        }
    }
}

template <int NG, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_bf16_small_m_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using T = SmallM<NG>;
    constexpr int NF = T::NF, D = T::D, STAGE = T::STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, kh = wave >> 2;

    // Blocks b and b + 8 run on the same XCD: the tiles_m row tiles of one column tile are neighbours there, so
    // the second one finds the weight rows in that XCD's L2.
    const int bid = blockIdx.x % p.n_tiles, split = blockIdx.x / p.n_tiles;
    const int in_xcd = bid >> 3;
    const int tm = in_xcd % p.tiles_m, tn = (in_xcd / p.tiles_m) * 8 + (bid & 7);
    if (tn >= p.tiles_n) return;
    const int nk = p.K / BK / p.splits;
    const size_t k_first = (size_t)split * nk * BK;
    const int m0 = tm * SM_BM, n0 = tn * T::BNS;

    // ---- DMA sources (row clamped into the matrix, chunk pre-swizzled by the row inside its region) ----
    // The kh = 0 waves move the activation rows (24 pieces, 6 a wave), the kh = 1 waves the weight rows
    // (BN / 8 pieces, BN / 32 a wave): loads return to a wave in order, so a wave that mixed the two would hold
    // its cache-resident activation rows back behind weight rows that come from HBM.
    constexpr int NWP = T::B_PIECES / 4;
    constexpr int NSRC = NWP > 6 ? NWP : 6;
    const bf16_t* src[NSRC];
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const int row = (wm + 4 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);               // 256-row blocks in the tile-major layout
        const bf16_t* wp = p.W + (size_t)(n >> 8) * p.w_tile_stride + (size_t)(n & 255) * p.w_row_stride + (k_first / BK) * p.w_kstride + chunk * 8;
        const bf16_t* ap = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + k_first + chunk * 8;
        src[i] = kh == 0 ? ap : wp;
    }

    auto issue = [&](int buf, int tt) {
        const int kt = min(tt, nk - 1);                     // past the end: the last tile again, into a free buffer
        unsigned char* base = smem + buf * STAGE;
        if (kh == 0) {
            if (LLADA_SM_DEBUG & 1) return;
#pragma unroll
            for (int i = 0; i < 6; ++i) glds16(src[i] + (size_t)kt * BK, base + (wm + 4 * i) * 1024);
        } else {
            if (LLADA_SM_DEBUG & 2) return;
            const size_t ks = (size_t)kt * p.w_kstride;
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                if constexpr (LLADA_SM_W_NT)
                    glds16_nt(src[i] + ks, base + SM_A_BYTES + (wm + 4 * i) * 1024);
                else
                    glds16(src[i] + ks, base + SM_A_BYTES + (wm + 4 * i) * 1024);
            }
        }
    };
    auto wait_landed = [&]() {   // own share of the oldest stage in flight has landed (D - 2 younger ones may not have)
        if (kh == 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LLADA_SM_DEBUG & 1) ? 0 : (D - 2) * 6) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LLADA_SM_DEBUG & 2) ? 0 : (D - 2) * NWP) : "memory");
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int coff = ((kh * 4 + fq) ^ (fr >> 1)) * 16;
    const int a_off = (wm * 48 + fr) * ROW_BYTES + coff;
    const int b_off = SM_A_BYTES + fr * ROW_BYTES + coff;

    f32x4 acc[3][NF];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s, s);

    int buf = 0, fill = D - 1;
    for (int t = 0; t < nk; ++t) {
        // own share of tile t has landed (the D - 2 younger stages may still be in flight); after the barrier
        // everyone's has, and everyone is done reading tile t - 1, whose buffer is refilled next
        wait_landed();
        SM_STAMP(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SM_STAMP(1);
        // the kh = 0 wave of every SIMD issues its refill before its MFMAs, the kh = 1 wave after them, so that one
        // of the two is free to feed the matrix pipe while the other's DMA instructions queue up
        if (kh == 0) issue(fill, t + D - 1);
        __builtin_amdgcn_sched_barrier(0);
        SM_STAMP(2);
        const unsigned char* sb = smem + buf * STAGE;
        bf16x8 a[3], b[NF];
        if (LLADA_SM_DEBUG & 4) {
            buf = (buf + 1 == D) ? 0 : buf + 1;
            fill = (fill + 1 == D) ? 0 : fill + 1;
            if (kh == 1) issue(fill, t + D - 1);
            continue;
        }
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) a[mf] = *(const bf16x8*)(sb + a_off + mf * 16 * ROW_BYTES);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) b[nf] = *(const bf16x8*)(sb + b_off + nf * 16 * ROW_BYTES);
        __builtin_amdgcn_sched_barrier(0);   // all reads in flight before the first MFMA waits for its operands
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int mf = 0; mf < 3; ++mf)
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf], a[mf], acc[mf][nf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        SM_STAMP(3);
        if (kh == 1) issue(fill, t + D - 1);
        buf = (buf + 1 == D) ? 0 : buf + 1;
        fill = (fill + 1 == D) ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing refills must land before the LDS is reused
    __syncthreads();
#if LLADA_SM_TRACE
    if (blockIdx.x == 0 && p.slab)
        for (int i = tid; i < 2048; i += NTHREADS) ((unsigned*)p.slab)[i] = ((const unsigned*)(smem + T::LDS_BYTES))[i];
#endif

    // ---- the two K halves of a row block: kh = 1 hands its sums over, kh = 0 adds them to its own ----
    unsigned char* mine = smem + wm * (3 * NF * 1024);
    if (kh == 1) {
#pragma unroll
        for (int mf = 0; mf < 3; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) *(f32x4*)(mine + ((mf * NF + nf) * 64 + lane) * 16) = acc[mf][nf];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int mf = 0; mf < 3; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const f32x4 o = *(const f32x4*)(mine + ((mf * NF + nf) * 64 + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mf][nf][e] += o[e];
        }
    store_rows48<NG, EPI>(p, acc, m0 + wm * 48, n0, mine, lane, split);
}
