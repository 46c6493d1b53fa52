// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] @ W[N,K]^T, float32 accumulation.
//
// Replaces the cuBLAS calls behind every nn.Linear of the mask predictor that the reference
// invokes at generate.py:88,92 and get_log_likelihood.py:38 (q/k/v, attn_out, ff_proj/up_proj,
// ff_out and the LM head).
//
// Tile: (MFR*32) x 256 x 64 per 512-thread workgroup (8 waves as 2(M) x 4(N), one workgroup per
// CU), v_mfma_f32_16x16x32_bf16, operands staged global -> LDS by 16-byte LDS-DMA
// (global_load_lds_dwordx4) into two stages.  LDS rows are 128 B (64 k); the 16-byte chunk c of
// row r is stored at chunk c ^ ((r >> 1) & 7), applied on the global SOURCE address (the DMA
// writes LDS linearly) and again on the ds_read_b128 address, which makes every 16-lane read
// group hit 16 distinct 16-byte slots.  The weight fragment is the MFMA A operand, so a lane
// ends up with 4 consecutive output columns of one row (8-byte stores, and rotary / SwiGLU
// partners in the same lane thanks to the packed weight layouts).
#include "common.h"

namespace {

constexpr int BN = 256;
constexpr int BK = 64;
constexpr int NTHREADS = 512;
constexpr int ROW_BYTES = BK * 2;  // 128
constexpr int EPI_PARTIAL = 100;   // internal: float32 partial tile of a split-K launch
// Diagnostic builds (-DLLADA_GEMM_DEBUG=bits, results are garbage): the pipelined loop without 1 its DMA after the first
// two K tiles, 2 its LDS fragment reads, 4 its MFMAs, 8 / 16 the DMA of the activation / weight rows only.
#ifndef LLADA_GEMM_DEBUG
#define LLADA_GEMM_DEBUG 0
#endif
// Issue priority of the MFMA blocks (timing experiments): 1 = raised around every block (the shipped form), 0 = never,
// 2 = waves 4-7 raised once before the loop and no flips.
#ifndef LLADA_GEMM_PRIO
#define LLADA_GEMM_PRIO 1
#endif

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    bf16_t* C;
    const bf16_t* R;
    const float* rope_cos;
    const float* rope_sin;
    int M, N, K;
    int lda, ldc, ldr;
    int tiles_m, tiles_n;
    int seq_len;
    const int* positions;  // [M] rotary table row of every token, or nullptr: row = token index % seq_len
    int rope_cols;  // columns < rope_cols get the rotary embedding (2 * d_model)
    float* slab;    // split-K partial sums [splits][M][N] float32, or nullptr
    int splits;     // K is cut into `splits` equal ranges, one workgroup each per output tile
    int n_tiles;    // tiles_m * tiles_n
    int no_store;   // diagnostic: skip the epilogue stores (timing experiments only)
    int plan_nfh;   // host only: 0 = choose in launch(); 1 / 2 = the planner's N tile (128 / 256 columns) ...
    int plan_splits;  // ... and its split count
    int plan_ng;    // host only: > 0 = the small-M loop (gemm_small_m.h) with 32 * plan_ng tile columns
    int band_m;     // M tiles per band of the tile walk
    int remap;      // 1: XCD-aware block -> tile map
    // RMSNorm of the updated residual stream fused into the split-K second pass (EPI_RESIDUAL only):
    const bf16_t* norm_w;   // [N] norm weight, or nullptr
    bf16_t* norm_out;       // [M, N] normalised rows
    float norm_eps;
    int* norm_fused;        // host flag: set to 1 when the second pass produced norm_out
    // Weight addressing: element (n0 + r, k0 + c) of W, n0 = tn * 256, k0 = kt * 64, is at
    //   W + tn * w_tile_stride + r * w_row_stride + kt * w_kstride + c.
    // Row-major [N, K]: (256 K, K, 64).  Tile-major (each 256 x 64 block contiguous, blocks of one
    // 256-row band consecutive along K): (256 K, 64, 256 * 64).
    size_t w_tile_stride;
    int w_row_stride, w_kstride;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Rotary table row of output row m (clamped into the matrix): the token's index inside its sequence, or the
// caller's own position (left-padded batches: the first real token of every row is position 0), clamped into
// the table.
__device__ __forceinline__ int rope_row(const GemmParams& p, int m) {
    const int mm = min(m, (int)p.M - 1);
    if (p.positions) return min(max(p.positions[mm], 0), p.seq_len - 1);
    return mm % p.seq_len;
}

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_uniform, 16, 0, 0);
}

// Non-temporal form (aux = 2) for weight rows that only one to three workgroups ever read (small-M GEMMs): the
// line is not kept in the caches on its way in.
__device__ __forceinline__ void glds16_nt(const bf16_t* src, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_uniform, 16, 0, 2);
}

// The same for lanes 0-31 only (512 contiguous LDS bytes).  EXEC is all ones in these kernels; the
// upper half is cleared around the one instruction and M0 is put back (the compiler tracks neither).
// M0 needs one wait state before an LDS-DMA.
__device__ __forceinline__ void glds16_lo32(const bf16_t* src, unsigned char* lds_uniform) {
    const unsigned lds_off = (unsigned)(size_t)(lds_ptr_t)lds_uniform;
    unsigned saved_m0;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 exec_hi, -1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(saved_m0)
        : "v"(src), "s"(lds_off)
        : "memory");
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
// tiles, and walk the tiles in bands of 8 M-tiles so the 32 tiles resident on an XCD share
// their A rows and W rows through that XCD's L2.  Bijective for any tile count.
__device__ __forceinline__ void tile_of_block(int bid, int nt, int tiles_m, int tiles_n, int& tm, int& tn,
                                              int GM = 8, int remap = 1) {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int wg = remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
    const int band = wg / (GM * tiles_n);
    const int first_m = band * GM;
    const int rows = min(GM, tiles_m - first_m);
    const int in_band = wg - band * GM * tiles_n;
    tm = first_m + in_band % rows;
    tn = in_band / rows;
}

// Epilogue shared by both main loops.  After the MFMAs a lane holds C[m][n .. n+3] for 4*MFR
// (row, column-group) pairs, i.e. 8-byte pieces scattered over 16 rows per fragment; stored like
// that every wave instruction touches 16 rows x 32 B and the tile's write-out costs ~10 us.  So the
// results (after the fused rotary / SwiGLU arithmetic, rounded to bf16) go through the wave's own
// slice of the now idle staging LDS (row stride = row bytes + 16, conflict-free for the 8-byte
// writes) and leave as 16 bytes per lane, 8 lanes per 128-byte row: whole cache lines.  The
// residual is read with the same full-line pattern during the write-out.
// The caller guarantees that no DMA or LDS read of the main loop is outstanding (barrier).
// MGP = m fragments staged per pass (0: as many as the idle pipeline stages hold); `split` = which
// float32 slab a partial tile goes to.
template <int MFR, int EPI, int MGP = 0, int NFH = 2>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[MFR][2 * NFH], int m0, int n0, int wm, int wn,
                                           int fr, int fq, unsigned char* smem, int wave, int lane, int split) {
    if (p.no_store == 1) return;
    constexpr int WCOLS = 32 * NFH;                    // output columns (before SwiGLU) of one wave
    constexpr int NF = 2 * NFH;                        // 16-column fragments per wave
    if constexpr (EPI == EPI_PARTIAL) {
        const int m_base = m0 + wm * MFR * 16 + fr;
        const int n_base = n0 + wn * WCOLS + fq * 4;
        float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mf = 0; mf < MFR; ++mf) {
            const int m = m_base + mf * 16;
            if (m >= p.M) continue;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n_base + nf * 16;
                if (n >= p.N) continue;
                *(f32x4*)(slab + (size_t)m * p.N + n) = acc[mf][nf];
            }
        }
    } else {
        constexpr bool SWI = (EPI == LLADA_EPI_SWIGLU);
        constexpr int OUTC = SWI ? WCOLS / 2 : WCOLS;  // output columns of this wave
        constexpr int ROWB = OUTC * 2 + 16;            // LDS row stride in bytes
        constexpr int MG = MGP ? MGP : ((MFR == 8) ? 4 : MFR);   // m fragments per pass (LDS budget)
        constexpr int LPR = OUTC * 2 / 16;             // lanes per output row in the write-out
        constexpr int RPI = 64 / LPR;                  // rows per write-out instruction
        unsigned char* my = smem + wave * (MG * 16 * ROWB);
        const int out_n0 = SWI ? (n0 + wn * WCOLS) / 2 : n0 + wn * WCOLS;   // first output column of the wave
        const int out_cols = SWI ? p.N / 2 : p.N;
        [[maybe_unused]] const bool rope = n0 < p.rope_cols;          // uniform per tile (rope_cols % 256 == 0)
        constexpr int NI = MG * 16 / RPI;              // write-out instructions per pass
        const int r_in = lane / LPR, chunk = lane % LPR;
        const int n = out_n0 + chunk * 8;
        const int n_ld = min(n, out_cols - 8);         // in-bounds column for unconditional loads
#pragma unroll
        for (int pass = 0; pass < MFR / MG; ++pass) {
            // All global loads of a pass (residual rows, rotary tables) are issued up front and
            // unconditionally (addresses clamped into the matrix), so they travel together: loaded
            // one by one behind a bounds branch, each costs a full L2 round trip (12 in a row).
            [[maybe_unused]] u32x4 resid[NI];
            if constexpr (EPI == LLADA_EPI_RESIDUAL) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int m = m0 + wm * MFR * 16 + pass * MG * 16 + i * RPI + r_in;
                    resid[i] = *(const u32x4*)(p.R + (size_t)min(m, (int)p.M - 1) * p.ldr + n_ld);
                }
            }
            [[maybe_unused]] f32x4 rc[MG][NFH], rs[MG][NFH];
            if constexpr (EPI == LLADA_EPI_ROPE_QKV) {
                if (rope) {
#pragma unroll
                    for (int ml = 0; ml < MG; ++ml) {
                        const int m = m0 + wm * MFR * 16 + (pass * MG + ml) * 16 + fr;
                        const int pos = rope_row(p, m);
#pragma unroll
                        for (int pr = 0; pr < NFH; ++pr) {
                            // rotary frequency of element 0: 32-column group of the packed 128-column head
                            const int j = (((n0 + wn * WCOLS + pr * 32) & 127) >> 5) * 16 + fq * 4;
                            rc[ml][pr] = *(const f32x4*)(p.rope_cos + (size_t)pos * 64 + j);
                            rs[ml][pr] = *(const f32x4*)(p.rope_sin + (size_t)pos * 64 + j);
                        }
                    }
                }
            }
            // ---- registers -> LDS (bf16) ----
#pragma unroll
            for (int ml = 0; ml < MG; ++ml) {
                const int mf = pass * MG + ml;
                unsigned char* row = my + (ml * 16 + fr) * ROWB + fq * 8;
                if constexpr (EPI == LLADA_EPI_PLAIN || EPI == LLADA_EPI_RESIDUAL) {
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) {
                        u32x2 o;
                        o[0] = pack_bf16x2(acc[mf][nf][0], acc[mf][nf][1]);
                        o[1] = pack_bf16x2(acc[mf][nf][2], acc[mf][nf][3]);
                        *(u32x2*)(row + nf * 32) = o;
                    }
                } else if constexpr (EPI == LLADA_EPI_ROPE_QKV) {
                    // packed head layout: 16-column blocks 0-15 | 64-79 | 16-31 | 80-95 | ...: fragments
                    // (2pr, 2pr+1) are rotary partners.  Columns >= rope_cols (V) pass through.
#pragma unroll
                    for (int pr = 0; pr < NFH; ++pr) {
                        const f32x4 x1 = acc[mf][2 * pr], x2 = acc[mf][2 * pr + 1];
                        f32x4 o1 = x1, o2 = x2;
                        if (rope) {
                            const f32x4 c = rc[ml][pr], sn = rs[ml][pr];
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
                    for (int pr = 0; pr < NFH; ++pr) {
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
            // ---- LDS -> global, whole rows ----
            // Projection outputs (tens of MB, read next by another kernel) leave with the
            // non-temporal hint; the residual stream (25 MB, re-read at once by the norm) stays an
            // ordinary store.
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rl = i * RPI + r_in;
                const int m = m0 + wm * MFR * 16 + pass * MG * 16 + rl;
                u32x4 v = *(const u32x4*)(my + rl * ROWB + chunk * 16);
                if constexpr (EPI == LLADA_EPI_RESIDUAL) {
                    const u32x4 r = resid[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = bf16_to_f32(r[e] & 0xFFFF) + bf16_to_f32(v[e] & 0xFFFF);
                        const float hi = bf16_to_f32(r[e] >> 16) + bf16_to_f32(v[e] >> 16);
                        v[e] = pack_bf16x2(lo, hi);
                    }
                }
                if (m < p.M && n < out_cols && p.no_store != 2) {
                    u32x4* dst = (u32x4*)(p.C + (size_t)m * p.ldc + n);
                    if constexpr (EPI == LLADA_EPI_RESIDUAL)
                        *dst = v;
                    else
                        __builtin_nontemporal_store(v, dst);
                }
            }
        }
    }
}

template <int MFR, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_bf16_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = MFR * 32;
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_LOADS = BM / 64;  // 16-byte DMA instructions per wave per stage
    constexpr int B_LOADS = BN / 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    tile_of_block(blockIdx.x % p.n_tiles, p.n_tiles, p.tiles_m, p.tiles_n, tm, tn, p.band_m, p.remap);
    const int nk = p.K / BK / p.splits;                         // K tiles of this workgroup's range
    const size_t k_first = (size_t)(blockIdx.x / p.n_tiles) * nk * BK;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane DMA source addresses (row clamped into the matrix; chunk pre-swizzled) ----
    const bf16_t* a_src[A_LOADS];
    const bf16_t* b_src[B_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int grow = min(m0 + row, p.M - 1);
        a_src[i] = p.A + (size_t)grow * p.lda + k_first + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int trow = min(n0 + row, p.N - 1) - n0;
        b_src[i] = p.W + (size_t)tn * p.w_tile_stride + (size_t)trow * p.w_row_stride +
                   (k_first / BK) * p.w_kstride + chunk * 8;
    }

    auto stage = [&](int s, int kt) {
        unsigned char* base = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) glds16(a_src[i] + kt * BK, base + (i * 8 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) glds16(b_src[i] + (size_t)kt * p.w_kstride, base + A_BYTES + (i * 8 + wave) * 1024);
    };

    f32x4 acc[MFR][4];
#pragma unroll
    for (int i = 0; i < MFR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = fr >> 1;
    const int a_row_off = (wm * MFR * 16 + fr) * ROW_BYTES;
    const int b_row_off = (wn * 64 + fr) * ROW_BYTES;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // stage kt has landed (vmcnt(0) + barrier); everyone is done with the other stage
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const unsigned char* As = smem + (kt & 1) * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ swz) * 16;
            bf16x8 af[MFR], wf[4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) wf[nf] = *(const bf16x8*)(Bs + b_row_off + nf * 16 * ROW_BYTES + coff);
#pragma unroll
            for (int mf = 0; mf < MFR; ++mf) af[mf] = *(const bf16x8*)(As + a_row_off + mf * 16 * ROW_BYTES + coff);
#pragma unroll
            for (int mf = 0; mf < MFR; ++mf)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], af[mf], acc[mf][nf], 0, 0, 0);
        }
    }

    __syncthreads();  // all waves are done reading the last stage: the LDS is free
    store_tile<MFR, EPI>(p, acc, m0, n0, wm, wn, fr, fq, smem, wave, lane, blockIdx.x / p.n_tiles);
}

// ------------------------------------------------------------------------------------------------
// Main kernel: the same tile, but the K loop is software-pipelined in quarter-tile phases so that
// no wave ever drains its LDS-DMA queue.
//
// Each 64-deep K tile is staged as four regions (A half 0/1 = the lower/upper half of every wave's
// rows, B half 0/1 = the lower/upper 32 of every wave's 64 columns) and consumed in four phases,
// one output quadrant each: (A0,B0) (A1,B0) (A1,B1) (A0,B1).  A phase
//   1. re-issues the DMA of the region whose fragments were read in the previous phase, for the
//      tile two steps ahead (so every region has ~7 phases = 1.75 K tiles to land),
//   2. reads from LDS the fragments the NEXT phase needs (register sets X/Y for the A halves, P/Q
//      for the B halves; A0 alternates between X and Y from tile to tile, hence two tiles per loop
//      iteration),
//   3. issues its quadrant's MFMAs from registers read one phase earlier,
//   4. waits for its own LDS reads (lgkmcnt(0)) and, with a COUNTED vmcnt, for its share of the
//      region the next phase reads, then joins the workgroup barrier (raw s_barrier: no drain).
// Data staged by DMA is therefore read one phase after the wait + barrier that retire it, and a
// region is refilled only after the barrier that follows its last read.
// The vmcnt counts assume every phase issues its region, so past the last K tile the phases keep
// issuing loads (of tile nk-1, into regions nobody reads again).
template <int MFR, int EPI, int NFH = 2, int WNT = 0>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_bf16_pipelined_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = MFR * 32;
    constexpr int HA = MFR / 2;                 // m fragments per A half (per wave)
    constexpr int AH_ROWS = MFR * 16;           // rows of one A-half region (both wave rows)
    constexpr int AH_BYTES = AH_ROWS * ROW_BYTES;
    constexpr int BNT = 128 * NFH;              // weight rows (output columns) of the tile: 256, or 128 for small M
    constexpr int BH_COLS = 16 * NFH;           // columns of one wave in one B half
    constexpr int BH_BYTES = 4 * BH_COLS * ROW_BYTES;   // 4 waves x 32 (16) columns
    constexpr int STAGE = 2 * AH_BYTES + 2 * BH_BYTES;
    constexpr int A_GROUPS = AH_ROWS / 8;       // 1-KiB DMA pieces per A-half region: 8, 12 or 16
    constexpr int NB = NFH;                     // pieces per wave per B-half region

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Every wave moves the same number of pieces per region, so the counted waits are compile-time
    // constants and the loop has no branches (a wave-uniform "does this wave move a second piece"
    // test costs 7-9 % of the kernel).  An A region of the 192-row tile is 12 KiB = 8 whole pieces
    // plus 8 half pieces (4 rows, lanes 0-31 of a second instruction issued with EXEC[63:32] = 0).
    constexpr bool two_a = A_GROUPS > 8;
    constexpr bool half_a = A_GROUPS == 12;
    // Non-temporal weight DMA (WNT = 1): a launch decision of the host, taken when M is so small that a weight
    // line is read by a handful of workgroups at most (weights_stream_once).  An HBM stream by LDS-DMA reads
    // 6.9-7.0 TB/s with it against 5.9 TB/s without; where ten or more workgroups share a line through the L2 it
    // would throw that reuse away.
    constexpr bool W_NT = WNT != 0;

    int tm, tn;
    tile_of_block(blockIdx.x % p.n_tiles, p.n_tiles, p.tiles_m, p.tiles_n, tm, tn, p.band_m, p.remap);
    const int nk = p.K / BK / p.splits;                         // K tiles of this workgroup's range
    const size_t k_first = (size_t)(blockIdx.x / p.n_tiles) * nk * BK;
    const int m0 = tm * BM, n0 = tn * BNT;

    // ---- DMA sources: region-local row rho -> tile row; chunk pre-swizzled by the local row ----
    const bf16_t* a_src[2][2];
    const bf16_t* b_src[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            {
                // second instruction: rows 64 + 4*wave .. +3 (half pieces) or piece wave + 8 (256-row tile)
                const int rho = (u == 0) ? wave * 8 + (lane >> 3)
                                         : (half_a ? 64 + wave * 4 + ((lane & 31) >> 3) : min(wave + 8, A_GROUPS - 1) * 8 + (lane >> 3));
                const int trow = (rho / (MFR * 8)) * (MFR * 16) + h * (MFR * 8) + rho % (MFR * 8);
                const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
                a_src[h][u] = p.A + (size_t)min(m0 + trow, p.M - 1) * p.lda + k_first + chunk * 8;
            }
            if (u < NB) {
                const int g = wave + 8 * u;
                const int rho = g * 8 + (lane >> 3);
                const int tcol = (rho / BH_COLS) * (2 * BH_COLS) + h * BH_COLS + rho % BH_COLS;
                const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
                const int n = min(n0 + tcol, p.N - 1);          // weight row; 256-row blocks in the tile-major layout
                b_src[h][u] = p.W + (size_t)(n >> 8) * p.w_tile_stride + (size_t)(n & 255) * p.w_row_stride +
                              (k_first / BK) * p.w_kstride + chunk * 8;
            }
        }

    auto issue_a = [&](int h, int tt) {
        if ((LLADA_GEMM_DEBUG & (1 | 8)) && tt >= 2) return;
        const int ks = min(tt, nk - 1) * BK;
        unsigned char* base = smem + (tt & 1) * STAGE + h * AH_BYTES;
        glds16(a_src[h][0] + ks, base + wave * 1024);
        if constexpr (half_a)
            glds16_lo32(a_src[h][1] + ks, base + 8 * 1024 + wave * 512);
        else if constexpr (two_a)
            glds16(a_src[h][1] + ks, base + (wave + 8) * 1024);
    };
    auto issue_b = [&](int h, int tt) {
        if ((LLADA_GEMM_DEBUG & (1 | 16)) && tt >= 2) return;
        const size_t ks = (size_t)min(tt, nk - 1) * p.w_kstride;
        unsigned char* base = smem + (tt & 1) * STAGE + 2 * AH_BYTES + h * BH_BYTES;
        if constexpr (W_NT) {
            glds16_nt(b_src[h][0] + ks, base + wave * 1024);
            if constexpr (NB == 2) glds16_nt(b_src[h][1] + ks, base + (wave + 8) * 1024);
        } else {
            glds16(b_src[h][0] + ks, base + wave * 1024);
            if constexpr (NB == 2) glds16(b_src[h][1] + ks, base + (wave + 8) * 1024);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = fr >> 1;
    const int c0 = ((0 + fq) ^ swz) * 16, c1 = ((4 + fq) ^ swz) * 16;
    const int a_off = (wm * (MFR * 8) + fr) * ROW_BYTES;
    const int b_off = 2 * AH_BYTES + (wn * BH_COLS + fr) * ROW_BYTES;

    auto read_a = [&](bf16x8 (&r)[HA][2], int h, int tt) {
        if ((LLADA_GEMM_DEBUG & 2) && tt >= 2) return;
        const unsigned char* base = smem + (tt & 1) * STAGE + h * AH_BYTES + a_off;
#pragma unroll
        for (int mf = 0; mf < HA; ++mf) {
            r[mf][0] = *(const bf16x8*)(base + mf * 16 * ROW_BYTES + c0);
            r[mf][1] = *(const bf16x8*)(base + mf * 16 * ROW_BYTES + c1);
        }
    };
    auto read_b = [&](bf16x8 (&r)[NFH][2], int h, int tt) {
        if ((LLADA_GEMM_DEBUG & 2) && tt >= 2) return;
        const unsigned char* base = smem + (tt & 1) * STAGE + h * BH_BYTES + b_off;
#pragma unroll
        for (int nf = 0; nf < NFH; ++nf) {
            r[nf][0] = *(const bf16x8*)(base + nf * 16 * ROW_BYTES + c0);
            r[nf][1] = *(const bf16x8*)(base + nf * 16 * ROW_BYTES + c1);
        }
    };

    f32x4 acc[MFR][2 * NFH];
#pragma unroll
    for (int i = 0; i < MFR; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NFH; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const bf16x8 (&a)[HA][2], const bf16x8 (&b)[NFH][2], int ha, int hb) {
        if (LLADA_GEMM_DEBUG & 4) return;
        if (LLADA_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mf = 0; mf < HA; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFH; ++nf)
                    acc[ha * HA + mf][hb * NFH + nf] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][ks], a[mf][ks], acc[ha * HA + mf][hb * NFH + nf], 0, 0, 0);
        if (LLADA_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };

    // A phase = [refill + LDS reads] barrier [MFMAs] barrier.  Waves 4-7 (the upper half of the tile
    // rows, which share their SIMDs with waves 0-3) run one barrier behind waves 0-3, so between two
    // barriers one wave of every SIMD issues MFMAs while its partner issues its DMA and LDS reads.
    // Before the second barrier a wave waits for its own LDS reads and, with a COUNTED vmcnt, for
    // its share of the region that is read two phases on (the other half confirms its share one
    // interval later, and only then may anyone read it): two B and two A regions are younger.
    // A region is refilled two phases after it was read, i.e. after both halves have confirmed
    // their reads of it.
#define WAIT_END(n_b, n_a)                                                                            \
    do {                                                                                              \
        if (two_a)                                                                                    \
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((n_b) * NB + (n_a) * 2) : "memory");  \
        else                                                                                          \
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((n_b) * NB + (n_a) * 1) : "memory");  \
    } while (0)
#define BARRIER()                           \
    do {                                    \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
#define PHASE(issue_stmt, read_stmt, mma_stmt)  \
    do {                                        \
        issue_stmt;                             \
        read_stmt;                              \
        BARRIER();                              \
        mma_stmt;                               \
        WAIT_END(2, 2);                         \
        BARRIER();                              \
    } while (0)

    bf16x8 X[HA][2], Y[HA][2], P[NFH][2], Q[NFH][2];

    // ---- prologue: tiles 0 and 1 in flight, in the order B0 A0 A1 B1 ----
    issue_b(0, 0);
    issue_a(0, 0);
    issue_a(1, 0);
    issue_b(1, 0);
    issue_b(0, 1);
    issue_a(0, 1);
    issue_a(1, 1);
    issue_b(1, 1);
    WAIT_END(3, 3);         // B0(0), A0(0) landed: A1(0) B1(0) B0(1) A0(1) A1(1) B1(1) are younger
    BARRIER();
    read_a(X, 0, 0);
    read_b(P, 0, 0);
    WAIT_END(2, 2);         // own reads done; A1(0), B1(0) landed: B0(1) A0(1) A1(1) B1(1) are younger
    BARRIER();
    if (wm == 1) BARRIER(); // stagger: the upper half runs one barrier behind
    if (LLADA_GEMM_PRIO == 2 && wm == 1) __builtin_amdgcn_s_setprio(1);

    for (int t = 0; t < nk; t += 2) {
        // ---- even tile t: A0 in X, A1 -> Y ----
        PHASE(issue_b(0, t + 2), read_a(Y, 1, t),     mma(X, P, 0, 0));
        PHASE(issue_a(0, t + 2), read_b(Q, 1, t),     mma(Y, P, 1, 0));
        PHASE(issue_a(1, t + 2), read_b(P, 0, t + 1), mma(Y, Q, 1, 1));
        PHASE(issue_b(1, t + 2), read_a(Y, 0, t + 1), mma(X, Q, 0, 1));
        // ---- odd tile t+1: A0 in Y, A1 -> X ----
        PHASE(issue_b(0, t + 3), read_a(X, 1, t + 1), mma(Y, P, 0, 0));
        PHASE(issue_a(0, t + 3), read_b(Q, 1, t + 1), mma(X, P, 1, 0));
        PHASE(issue_a(1, t + 3), read_b(P, 0, t + 2), mma(X, Q, 1, 1));
        PHASE(issue_b(1, t + 3), read_a(X, 0, t + 2), mma(Y, Q, 0, 1));
    }
    if (wm == 0) BARRIER(); // the lower half waits for the upper half's last interval
#undef WAIT_END
#undef PHASE
#undef BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing refills must land before the LDS is released

    __syncthreads();  // all waves are done reading the last stage: the LDS is free
    store_tile<MFR, EPI, 0, NFH>(p, acc, m0, n0, wm, wn, fr, fq, smem, wave, lane, blockIdx.x / p.n_tiles);
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the pipelined kernel: a workgroup walks the work items b, b+G, b+2G, ... (G =
// grid size, a multiple of 8, so all its items map to the XCD it runs on) as ONE continuous stream
// of K tiles.  The refills issued in an item's last two K tiles already fetch the first two K
// tiles of the next item, and the fragment reads run ahead in the same way, so the DMA pipeline
// never drains between output tiles: the only per-tile cost left is the write-out itself, which
// uses its own LDS slice next to the two pipeline stages (2 m-fragments per pass and wave) while
// the next item's operands are already landing.  Stores count in vmcnt like the DMA, so the first
// waits after a write-out are stricter than needed (never weaker).
template <int MFR, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_bf16_stream_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = MFR * 32;
    constexpr int HA = MFR / 2;
    constexpr int AH_ROWS = MFR * 16;
    constexpr int AH_BYTES = AH_ROWS * ROW_BYTES;
    constexpr int BH_BYTES = 128 * ROW_BYTES;
    constexpr int STAGE = 2 * AH_BYTES + 2 * BH_BYTES;
    constexpr int A_GROUPS = AH_ROWS / 8;
    constexpr int NB = 2;
    constexpr bool two_a = A_GROUPS > 8;
    constexpr bool half_a = A_GROUPS == 12;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    unsigned char* epi_lds = smem + 2 * STAGE;

    const int nk = p.K / BK / p.splits;
    const int n_items = p.n_tiles * p.splits;
    const int G = gridDim.x;
    int item = blockIdx.x;

    // ---- lane constants of the DMA sources: tile row and pre-swizzled chunk per (half, piece) ----
    int a_trow[2][2], a_chunk[2], b_tcol[2][2], b_chunk[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rho = (u == 0) ? wave * 8 + (lane >> 3)
                                 : (half_a ? 64 + wave * 4 + ((lane & 31) >> 3) : min(wave + 8, A_GROUPS - 1) * 8 + (lane >> 3));
        a_chunk[u] = ((lane & 7) ^ ((rho >> 1) & 7)) * 8;
        const int rb = (wave + 8 * u) * 8 + (lane >> 3);
        b_chunk[u] = ((lane & 7) ^ ((rb >> 1) & 7)) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            a_trow[h][u] = (rho / (MFR * 8)) * (MFR * 16) + h * (MFR * 8) + rho % (MFR * 8);
            b_tcol[h][u] = (rb >> 5) * 64 + h * 32 + (rb & 31);
        }
    }
    const bf16_t* a_src[2][2];
    const bf16_t* b_src[2][2];
    auto decode = [&](int it, int& m0, int& n0, int& split) {
        int tm, tn;
        tile_of_block(it % p.n_tiles, p.n_tiles, p.tiles_m, p.tiles_n, tm, tn, p.band_m, p.remap);
        m0 = tm * BM;
        n0 = tn * BN;
        split = it / p.n_tiles;
    };
    auto set_sources = [&](int m0, int n0, int split) {
        const size_t k_first = (size_t)split * nk * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a_src[h][u] = p.A + (size_t)min(m0 + a_trow[h][u], p.M - 1) * p.lda + k_first + a_chunk[u];
                b_src[h][u] = p.W + (size_t)(n0 / BN) * p.w_tile_stride +
                              (size_t)(min(n0 + b_tcol[h][u], p.N - 1) - n0) * p.w_row_stride +
                              (k_first / BK) * p.w_kstride + b_chunk[u];
            }
    };

    // kt = K tile of the item the sources point at; par = LDS stage (parity of the stream position)
    auto issue_a = [&](int h, int kt, int par) {
        const int ks = min(kt, nk - 1) * BK;
        unsigned char* base = smem + par * STAGE + h * AH_BYTES;
        glds16(a_src[h][0] + ks, base + wave * 1024);
        if constexpr (half_a)
            glds16_lo32(a_src[h][1] + ks, base + 8 * 1024 + wave * 512);
        else if constexpr (two_a)
            glds16(a_src[h][1] + ks, base + (wave + 8) * 1024);
    };
    auto issue_b = [&](int h, int kt, int par) {
        const size_t ks = (size_t)min(kt, nk - 1) * p.w_kstride;
        unsigned char* base = smem + par * STAGE + 2 * AH_BYTES + h * BH_BYTES;
        glds16(b_src[h][0] + ks, base + wave * 1024);
        glds16(b_src[h][1] + ks, base + (wave + 8) * 1024);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = fr >> 1;
    const int c0 = ((0 + fq) ^ swz) * 16, c1 = ((4 + fq) ^ swz) * 16;
    const int a_off = (wm * (MFR * 8) + fr) * ROW_BYTES;
    const int b_off = 2 * AH_BYTES + (wn * 32 + fr) * ROW_BYTES;

    auto read_a = [&](bf16x8 (&r)[HA][2], int h, int par) {
        const unsigned char* base = smem + par * STAGE + h * AH_BYTES + a_off;
#pragma unroll
        for (int mf = 0; mf < HA; ++mf) {
            r[mf][0] = *(const bf16x8*)(base + mf * 16 * ROW_BYTES + c0);
            r[mf][1] = *(const bf16x8*)(base + mf * 16 * ROW_BYTES + c1);
        }
    };
    auto read_b = [&](bf16x8 (&r)[2][2], int h, int par) {
        const unsigned char* base = smem + par * STAGE + h * BH_BYTES + b_off;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            r[nf][0] = *(const bf16x8*)(base + nf * 16 * ROW_BYTES + c0);
            r[nf][1] = *(const bf16x8*)(base + nf * 16 * ROW_BYTES + c1);
        }
    };

    f32x4 acc[MFR][4];
#pragma unroll
    for (int i = 0; i < MFR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const bf16x8 (&a)[HA][2], const bf16x8 (&b)[2][2], int ha, int hb) {
        if (LLADA_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mf = 0; mf < HA; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
                    acc[ha * HA + mf][hb * 2 + nf] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][ks], a[mf][ks], acc[ha * HA + mf][hb * 2 + nf], 0, 0, 0);
        if (LLADA_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };

#define WAIT_END(n_b, n_a) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((n_b) * NB + (n_a) * (two_a ? 2 : 1)) : "memory")
#define BARRIER()                           \
    do {                                    \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
#define PHASE(issue_stmt, read_stmt, mma_stmt)  \
    do {                                        \
        issue_stmt;                             \
        read_stmt;                              \
        BARRIER();                              \
        mma_stmt;                               \
        WAIT_END(2, 2);                         \
        BARRIER();                              \
    } while (0)

    bf16x8 X[HA][2], Y[HA][2], P[2][2], Q[2][2];

    int m0, n0, split;
    decode(item, m0, n0, split);
    set_sources(m0, n0, split);
    issue_b(0, 0, 0);
    issue_a(0, 0, 0);
    issue_a(1, 0, 0);
    issue_b(1, 0, 0);
    issue_b(0, 1, 1);
    issue_a(0, 1, 1);
    issue_a(1, 1, 1);
    issue_b(1, 1, 1);
    WAIT_END(3, 3);
    BARRIER();
    read_a(X, 0, 0);
    read_b(P, 0, 0);
    WAIT_END(2, 2);
    BARRIER();
    if (wm == 1) BARRIER();  // stagger: the upper half runs one barrier behind
    if (LLADA_GEMM_PRIO == 2 && wm == 1) __builtin_amdgcn_s_setprio(1);

    for (;;) {
        int ik = 2;          // next K tile to fetch, relative to the item the sources point at
        int nm0 = 0, nn0 = 0, nsplit = 0;
        bool more = false;
        for (int t = 0; t < nk; t += 2) {
            if (t == nk - 2) {   // the last two refills of this item fetch the head of the next one
                more = item + G < n_items;
                if (more) {
                    decode(item + G, nm0, nn0, nsplit);
                    set_sources(nm0, nn0, nsplit);
                    ik = 0;
                }
            }
            PHASE(issue_b(0, ik, 0), read_a(Y, 1, 0), mma(X, P, 0, 0));
            PHASE(issue_a(0, ik, 0), read_b(Q, 1, 0), mma(Y, P, 1, 0));
            PHASE(issue_a(1, ik, 0), read_b(P, 0, 1), mma(Y, Q, 1, 1));
            PHASE(issue_b(1, ik, 0), read_a(Y, 0, 1), mma(X, Q, 0, 1));
            PHASE(issue_b(0, ik + 1, 1), read_a(X, 1, 1), mma(Y, P, 0, 0));
            PHASE(issue_a(0, ik + 1, 1), read_b(Q, 1, 1), mma(X, P, 1, 0));
            PHASE(issue_a(1, ik + 1, 1), read_b(P, 0, 0), mma(X, Q, 1, 1));
            PHASE(issue_b(1, ik + 1, 1), read_a(X, 0, 0), mma(Y, Q, 0, 1));
            ik += 2;
        }
        store_tile<MFR, EPI, 2>(p, acc, m0, n0, wm, wn, fr, fq, epi_lds, wave, lane, split);
        if (!more) break;
#pragma unroll
        for (int i = 0; i < MFR; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        item += G;
        m0 = nm0;
        n0 = nn0;
        split = nsplit;
    }
    if (wm == 0) BARRIER();  // the lower half waits for the upper half's last interval
#undef WAIT_END
#undef PHASE
#undef BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing refills must land before the LDS is released
}

// Elementwise epilogue on float32 sums, identical to store_tile's: v1 = columns n1..n1+3 of row m,
// v2 = columns n1+16..n1+19 (the rotary / SwiGLU partners; unused otherwise).
template <int EPI>
__device__ __forceinline__ void finish_and_store(const GemmParams& p, int m, int n1, f32x4 v1, f32x4 v2) {
    u32x2 w1, w2;
    if constexpr (EPI == LLADA_EPI_PLAIN || EPI == LLADA_EPI_RESIDUAL) {
        if constexpr (EPI == LLADA_EPI_RESIDUAL) {
            const u32x2 r = *(const u32x2*)(p.R + (size_t)m * p.ldr + n1);
            v1[0] = bf16_to_f32(r[0] & 0xFFFF) + round_bf16(v1[0]);
            v1[1] = bf16_to_f32(r[0] >> 16) + round_bf16(v1[1]);
            v1[2] = bf16_to_f32(r[1] & 0xFFFF) + round_bf16(v1[2]);
            v1[3] = bf16_to_f32(r[1] >> 16) + round_bf16(v1[3]);
        }
        w1[0] = pack_bf16x2(v1[0], v1[1]);
        w1[1] = pack_bf16x2(v1[2], v1[3]);
        *(u32x2*)(p.C + (size_t)m * p.ldc + n1) = w1;
    } else if constexpr (EPI == LLADA_EPI_ROPE_QKV) {
        f32x4 o1 = v1, o2 = v2;
        if (n1 < p.rope_cols) {
            const int ch = n1 & 127;                       // column inside the packed head
            const int j = (ch >> 5) * 16 + (ch & 15);      // rotary frequency of element 0
            const int pos = rope_row(p, m);
            const f32x4 cs = *(const f32x4*)(p.rope_cos + (size_t)pos * 64 + j);
            const f32x4 sn = *(const f32x4*)(p.rope_sin + (size_t)pos * 64 + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = round_bf16(v1[e]), b = round_bf16(v2[e]);
                o1[e] = a * cs[e] - b * sn[e];
                o2[e] = b * cs[e] + a * sn[e];
            }
        }
        w1[0] = pack_bf16x2(o1[0], o1[1]);
        w1[1] = pack_bf16x2(o1[2], o1[3]);
        w2[0] = pack_bf16x2(o2[0], o2[1]);
        w2[1] = pack_bf16x2(o2[2], o2[3]);
        *(u32x2*)(p.C + (size_t)m * p.ldc + n1) = w1;
        *(u32x2*)(p.C + (size_t)m * p.ldc + n1 + 16) = w2;
    } else {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gb = round_bf16(v1[e]);
            const float act = round_bf16(silu_f32(gb));
            o[e] = act * round_bf16(v2[e]);
        }
        w1[0] = pack_bf16x2(o[0], o[1]);
        w1[1] = pack_bf16x2(o[2], o[3]);
        *(u32x2*)(p.C + (size_t)m * p.ldc + (n1 >> 5) * 16 + (n1 & 15)) = w1;
    }
}

// Second pass of a split-K GEMM: adds the float32 partial tiles in split order (deterministic) and
// applies the epilogue exactly as store_tile does.  One thread per 4 consecutive columns (PLAIN,
// RESIDUAL) or per rotary / SwiGLU partner pair of 4-column groups (columns n and n + 16).
template <int EPI>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmParams p) {
    constexpr bool PAIRED = (EPI == LLADA_EPI_ROPE_QKV || EPI == LLADA_EPI_SWIGLU);
    const int per_row = PAIRED ? p.N / 8 : p.N / 4;
    const int64_t total = (int64_t)p.M * per_row;
    const size_t slab_stride = (size_t)p.M * p.N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / per_row), c = (int)(i % per_row);
        const int n1 = PAIRED ? (c >> 2) * 32 + (c & 3) * 4 : c * 4;
        const float* src = p.slab + (size_t)m * p.N + n1;
        f32x4 v1 = *(const f32x4*)src, v2 = {0.f, 0.f, 0.f, 0.f};
        if (PAIRED) v2 = *(const f32x4*)(src + 16);
        // the partial tiles are fetched four at a time (independent loads in flight) and added in split order
        int sidx = 1;
        for (; sidx + 3 < p.splits; sidx += 4) {
            f32x4 t1[4], t2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t1[j] = *(const f32x4*)(src + (sidx + j) * slab_stride);
                if (PAIRED) t2[j] = *(const f32x4*)(src + (sidx + j) * slab_stride + 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v1[e] += t1[j][e];
                    if (PAIRED) v2[e] += t2[j][e];
                }
        }
        for (; sidx < p.splits; ++sidx) {
            const f32x4 t1 = *(const f32x4*)(src + sidx * slab_stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) v1[e] += t1[e];
            if (PAIRED) {
                const f32x4 t2 = *(const f32x4*)(src + sidx * slab_stride + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v2[e] += t2[e];
            }
        }
        finish_and_store<EPI>(p, m, n1, v1, v2);
    }
}

// Second pass of a split-K GEMM with the residual epilogue, fused with the RMSNorm that follows it in
// the model (attn_out -> ff_norm, ff_out -> next attn_norm / final norm): one workgroup per row adds
// the float32 partial tiles in split order, adds the residual, stores the new residual row and its
// normalised copy.  Same arithmetic and the same summation order as splitk_reduce_kernel<RESIDUAL>
// followed by rmsnorm_kernel (elementwise.hip), so the results are bit-identical to the two kernels.
template <int MAXV>
__global__ void __launch_bounds__(256) splitk_reduce_norm_kernel(const GemmParams p) {
    __shared__ float red[4];
    const int chunks = p.N / 8;
    const int tid = threadIdx.x;
    const size_t slab_stride = (size_t)p.M * p.N;
    for (int m = blockIdx.x; m < p.M; m += gridDim.x) {
        u16x8 v[MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + i * 256;
            if (c < chunks) {
                const float* src = p.slab + (size_t)m * p.N + c * 8;
                f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
                // four partial tiles per round: independent loads in flight, added in split order
                int sidx = 1;
                for (; sidx + 3 < p.splits; sidx += 4) {
                    f32x4 ta[4], tb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ta[j] = *(const f32x4*)(src + (sidx + j) * slab_stride);
                        tb[j] = *(const f32x4*)(src + (sidx + j) * slab_stride + 4);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] += ta[j][e];
                            b[e] += tb[j][e];
                        }
                }
                for (; sidx < p.splits; ++sidx) {
                    const f32x4 ta = *(const f32x4*)(src + sidx * slab_stride);
                    const f32x4 tb = *(const f32x4*)(src + sidx * slab_stride + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] += ta[e];
                        b[e] += tb[e];
                    }
                }
                const u16x8 r = *(const u16x8*)(p.R + (size_t)m * p.ldr + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[i][e] = f32_to_bf16(bf16_to_f32(r[e]) + round_bf16(a[e]));
                    v[i][4 + e] = f32_to_bf16(bf16_to_f32(r[4 + e]) + round_bf16(b[e]));
                }
                *(u16x8*)(p.C + (size_t)m * p.ldc + c * 8) = v[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf16_to_f32(v[i][e]);
                    ss += f * f;
                }
            }
        }
        ss = wave_reduce_sum(ss);
        __syncthreads();  // red[] reuse across rows
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float rs = rsqrtf(tot / (float)p.N + p.norm_eps);
        const u16x8* wp = (const u16x8*)p.norm_w;
        u16x8* op = (u16x8*)(p.norm_out + (size_t)m * p.N);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + i * 256;
            if (c < chunks) {
                const u16x8 wv = wp[c];
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xn = round_bf16(bf16_to_f32(v[i][e]) * rs);
                    o[e] = f32_to_bf16(xn * bf16_to_f32(wv[e]));
                }
                op[c] = o;
            }
        }
    }
}

#include "gemm_small_m.h"

int g_use_pipelined = 1;  // test hook: 0 forces the simple two-stage kernel, 2 the persistent stream kernel, 3 never it
int g_stream_grid = 256;  // workgroups of the persistent kernel (one per CU)
int g_band_m = 8, g_remap = 1;  // tuning hooks for the block -> tile walk
int g_force_splits = 0;   // test hook: > 0 forces the split count when a workspace is given
thread_local size_t g_slab_bytes = 0;  // capacity of the caller's split-K workspace for the current call
thread_local int g_last_plan[4] = {0, 0, 0, 0};   // test hook: {tile rows, tile columns, K ranges, persistent form} of the last main-kernel launch
inline void note_plan(int rows, int cols, int splits, int stream_form) {
    g_last_plan[0] = rows; g_last_plan[1] = cols; g_last_plan[2] = splits; g_last_plan[3] = stream_form;
}

// Split count for a launch that would leave CUs idle: estimated time = rounds * (fill/drain +
// K steps) + the reduce pass, in microseconds at ~1.25 us per K step.
int pick_splits(int nt, int nk_total, int64_t M, int N, bool pipelined_ok) {
    if (g_slab_bytes == 0) return 1;
    auto fits = [&](int S) { return (size_t)S * M * N * 4 <= g_slab_bytes; };
    if (g_force_splits > 0) return (nk_total % g_force_splits == 0 && fits(g_force_splits)) ? g_force_splits : 1;
    if (nt >= 192) return 1;
    int best = 1;
    double best_cost = ((nt + 255) / 256) * (8.0 + 1.25 * nk_total);
    for (int S = 2; S <= 16; ++S) {
        if (nk_total % S) continue;
        const int nk = nk_total / S;
        if (nk < 4 || (pipelined_ok && (nk & 1))) continue;
        if (!fits(S)) break;
        const double cost = ((nt * S + 255) / 256) * (8.0 + 1.25 * nk) + 4.0 + (double)S * M * N * 4 / 3.0e6;
        if (cost < best_cost) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}

template <int MFR, int EPI, int NFH = 2, int WNT = 0>
int launch_main(const GemmParams& q, int grid, hipStream_t stream) {
    constexpr int BM = MFR * 32;
    constexpr int lds = 2 * (BM + 128 * NFH) * ROW_BYTES;
    const int nk = q.K / BK / q.splits;
    if constexpr (NFH == 1) {   // 128-column tile: the pipelined loop only (the caller checked nk >= 4 and even)
        static OncePerDevice attr4_set;
        if (attr4_set.first()) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pipelined_kernel<MFR, EPI, 1, WNT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                llada_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
                return LLADA_ERR_HIP;
            }
        }
        note_plan(BM, 128, q.splits, 0);
        hipLaunchKernelGGL((gemm_bf16_pipelined_kernel<MFR, EPI, 1, WNT>), dim3(grid), dim3(NTHREADS), lds, stream, q);
        LLADA_CHECK_LAUNCH("gemm_bf16_pipelined_kernel");
        return LLADA_OK;
    } else {
    if constexpr (MFR != 8) {  // the persistent form of the 256-row tile spills registers
        // The persistent form pays off where a workgroup gets several tiles and the write-out issues no
        // loads of its own (gate/up projection: 6 tiles per CU at M = 3072); elsewhere it is neutral or
        // slower in situ.  Hook values: 2 = always, 3 = never.
        const bool stream_ok = WNT == 0 && (g_use_pipelined == 2 ||
                                            (g_use_pipelined == 1 && EPI == LLADA_EPI_SWIGLU && grid >= 2 * g_stream_grid));
        if (stream_ok && nk >= 4 && (nk & 1) == 0) {
            constexpr int stream_lds = lds + 8 * 2 * 16 * 144;   // + the write-out slices (2 m-fragments per wave)
            static OncePerDevice attr3_set;
            if (attr3_set.first()) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_stream_kernel<MFR, EPI>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, stream_lds);
                if (e != hipSuccess) {
                    llada_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
                    return LLADA_ERR_HIP;
                }
            }
            const int sgrid = grid < g_stream_grid ? grid : g_stream_grid;
            note_plan(BM, 256, q.splits, 1);
            hipLaunchKernelGGL((gemm_bf16_stream_kernel<MFR, EPI>), dim3(sgrid), dim3(NTHREADS), stream_lds, stream, q);
            LLADA_CHECK_LAUNCH("gemm_bf16_stream_kernel");
            return LLADA_OK;
        }
    }
    {
        if (g_use_pipelined && nk >= 4 && (nk & 1) == 0) {
            static OncePerDevice attr2_set;
            if (attr2_set.first()) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pipelined_kernel<MFR, EPI, 2, WNT>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) {
                    llada_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
                    return LLADA_ERR_HIP;
                }
            }
            note_plan(BM, 256, q.splits, 0);
            hipLaunchKernelGGL((gemm_bf16_pipelined_kernel<MFR, EPI, 2, WNT>), dim3(grid), dim3(NTHREADS), lds, stream, q);
            LLADA_CHECK_LAUNCH("gemm_bf16_pipelined_kernel");
            return LLADA_OK;
        }
    }
    static OncePerDevice attr_set;
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<MFR, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            llada_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
            return LLADA_ERR_HIP;
        }
    }
    note_plan(BM, 256, q.splits, 0);
    hipLaunchKernelGGL((gemm_bf16_kernel<MFR, EPI>), dim3(grid), dim3(NTHREADS), lds, stream, q);
    LLADA_CHECK_LAUNCH("gemm_bf16_kernel");
    return LLADA_OK;
    }
}

// Non-temporal weight DMA is worth it where (almost) nobody re-reads a weight line: at most three 192-row tile
// rows, i.e. one sequence of up to 576 tokens or the pruned last layer.  The 256-row tile never runs there.
template <int MFR, int EPI, int NFH>
int launch_main_nt(const GemmParams& q, int grid, hipStream_t stream) {
    if constexpr (MFR == 8) {
        return launch_main<MFR, EPI, NFH, 0>(q, grid, stream);
    } else {
        const bool weights_stream_once = g_use_pipelined == 1 && q.M <= 3 * 192;
        return weights_stream_once ? launch_main<MFR, EPI, NFH, 1>(q, grid, stream) : launch_main<MFR, EPI, NFH, 0>(q, grid, stream);
    }
}

int g_no_store = 0;       // diagnostic hook
int g_force_cols = 0;     // test hook: 128 / 256 forces the N tile, 0 = automatic
int g_force_small_m = 0;  // test hook: > 0 runs the loop of gemm_small_m.h with 32 * value tile columns (never chosen automatically:
                          // inside a denoising step it is 1-2 % slower than the plans below, see the header)

// ---- launch plans for under-filled launches ------------------------------------------------------------------
// At M = 384 the 256-column tile gives 96-192 work items for 256 CUs, so the tile height, the tile width (the
// pipelined loop also runs 128 columns wide) and the split count are chosen together.  The planner runs for every
// launch the legacy plan leaves under 192 tiles (one or two sequences, the pruned last layer, and the N = 4096
// projections up to M ~ 2000); its cost model is fitted to measured launches of the four LLaDA-8B projections at
// M = 384 (weights cold) and is extrapolated, not re-fitted, above that: microseconds =
// rounds * (8 + t_step * K steps) + second pass of a split (4 + slab bytes / 3 TB/s), with t_step per
// (tile rows, tile columns).  Model against measurement: QKV 192 x 128 unsplit 49.6 (49.6), 128 x 256 unsplit 56.6
// (58.1); attention output 192 x 128 in 4 ranges 30.8 (27.1); MLP output 192 x 256 in 8 ranges 50.4 (49.8).
struct Plan {
    int mfr, nfh, splits;
    int ng;   // > 0: the 192-row small-M loop with 32 * ng columns (mfr / nfh unused)
};

inline double small_m_step_us(int mfr, int nfh) {
    if (nfh == 2) return mfr == 4 ? 0.76 : (mfr == 6 ? 0.90 : 1.20);
    return mfr == 4 ? 0.50 : 0.65;
}

inline double plan_cost(int64_t M, int N, int nk_total, int mfr, int nfh, int S) {
    const int bm = mfr * 32, bn = 128 * nfh;
    const int64_t items = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * S;
    const double rounds = (double)((items + 255) / 256);
    // the K loops cannot finish before every weight byte has left HBM once (5.8 TB/s): binds at M <= 64, where 256
    // items of a few K steps each would otherwise look free (MLP output at M = 32: 16 ranges 29.7 us, 8 ranges of the
    // 128-column tile 26.5)
    const double loop = fmax(rounds * small_m_step_us(mfr, nfh) * (nk_total / S), (double)N * nk_total * BK * 2 / 5.8e6);
    double cost = rounds * 8.0 + loop;
    // a split plan must win by 5 %: inside a forward its partial tiles and second pass cost a few microseconds more
    // than in the isolated launches the model is fitted to (QKV at M = 192: 192 x 256 in 4 ranges 35.3 + 8.2 us in the
    // step against 37 for unsplit 128 x 128 tiles, which the model had 1 us behind)
    if (S > 1) cost = (cost + 4.0 + (double)S * M * N * 4 / 3.0e6) * 1.05;
    return cost;
}

// Joint choice for a launch the legacy plan (pick_mfr, then pick_splits) leaves under-filled; {0, 0, 0, 0} = keep
// the legacy plan (always at large M: its choices are tuned and do not move).
inline Plan choose_small_m_plan(int64_t M, int N, int K, int legacy_mfr) {
    const Plan keep = {0, 0, 0, 0};
    if (g_force_small_m > 0) {
        const int S = g_force_splits > 0 ? g_force_splits : 1;
        const bool ok = K % BK == 0 && (K / BK) % S == 0 && (S == 1 || (size_t)S * M * N * 4 <= g_slab_bytes);
        return ok ? Plan{0, 0, S, g_force_small_m} : keep;
    }
    if (g_use_pipelined != 1 || g_force_splits > 0 || g_force_cols != 0 || K % BK) return keep;
    const int nk_total = K / BK;
    const int lbm = legacy_mfr * 32;
    const int64_t legacy_tiles = ((M + lbm - 1) / lbm) * ((N + BN - 1) / BN);
    // pick_splits would not split 192 tiles or more: the legacy plan stands, except for a single round over at most
    // three tile rows (the gate/up projection of one sequence).  That one is searched like the others: at M = 384 the
    // search returns the legacy tile (192 x 256, unsplit; planned, i.e. with the non-temporal weight DMA), at
    // M <= 192 the 192 x 128 tile, whose single tile row has no padding rows (measured at M = 192: 51.8 us against
    // 56.2 for 128 x 256, which pads the second tile row to twice its 64 tokens).
    if (legacy_tiles >= 192 && !(legacy_tiles <= 256 && M <= 3 * 192)) return keep;
    Plan best = keep;
    double best_cost = 1e30;
    for (int mfr : {4, 6})
        for (int nfh : {2, 1}) {
            if (nfh == 1 && N % 128) continue;
            for (int S = 1; S <= 16; S *= 2) {
                if (nk_total % S) continue;
                const int nk = nk_total / S;
                if (nk < 4 || (nk & 1)) continue;
                if (S > 1 && (size_t)S * M * N * 4 > g_slab_bytes) break;
                const double c = plan_cost(M, N, nk_total, mfr, nfh, S);
                if (c < best_cost) {
                    best_cost = c;
                    best = Plan{mfr, nfh, S, 0};
                }
            }
        }
    return best;
}

// Test hook path (llada_gemm_force_tile_cols): split count for the 128-column tile, 0 = not applicable.
template <int MFR>
int pick_narrow(int64_t M, int N, int nk_total) {
    if (g_force_cols != 128 || N % 128 != 0) return 0;
    int best = 0;
    double best_cost = 1e30;
    for (int S = 1; S <= 16; S *= 2) {
        if (g_force_splits > 0 && S != g_force_splits) continue;
        if (nk_total % S) continue;
        const int nk = nk_total / S;
        if (nk < 4 || (nk & 1)) continue;
        if (S > 1 && (size_t)S * M * N * 4 > g_slab_bytes) continue;
        const double c = plan_cost(M, N, nk_total, MFR, 1, S);
        if (c < best_cost) {
            best_cost = c;
            best = S;
        }
    }
    return best;
}

template <int EPI>
int finish_split(const GemmParams& q, hipStream_t stream) {
    if constexpr (EPI == LLADA_EPI_RESIDUAL) {
        if (q.norm_out && q.norm_w && q.N % 8 == 0 && q.N <= 256 * 8 * 4) {
            const int rgrid = q.M < 8192 ? q.M : 8192;
            if (q.N <= 256 * 8 * 2)
                hipLaunchKernelGGL((splitk_reduce_norm_kernel<2>), dim3(rgrid), dim3(256), 0, stream, q);
            else
                hipLaunchKernelGGL((splitk_reduce_norm_kernel<4>), dim3(rgrid), dim3(256), 0, stream, q);
            LLADA_CHECK_LAUNCH("splitk_reduce_norm_kernel");
            if (q.norm_fused) *q.norm_fused = 1;
            return LLADA_OK;
        }
    }
    const int64_t items = (int64_t)q.M * q.N / 4;
    const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), dim3(grid), dim3(256), 0, stream, q);
    LLADA_CHECK_LAUNCH("splitk_reduce_kernel");
    return LLADA_OK;
}

inline void set_weight_strides(GemmParams& q, const GemmParams& p) {
    q.w_tile_stride = (size_t)BN * p.K;
    if (p.w_row_stride == BK) {   // tile-major weights (the entry point checked N % 256 == 0)
        q.w_kstride = BN * BK;
    } else {
        q.w_row_stride = p.K;
        q.w_kstride = BK;
    }
}

template <int NG, int EPI>
int launch_small_m_kernel(const GemmParams& q, int grid, hipStream_t stream) {
    static OncePerDevice attr_set;
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_small_m_kernel<NG, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SmallM<NG>::LDS_BYTES + LLADA_SM_TRACE * 8192);
        if (e != hipSuccess) {
            llada_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
            return LLADA_ERR_HIP;
        }
    }
    note_plan(SM_BM, 32 * NG, q.splits, 0);
    hipLaunchKernelGGL((gemm_bf16_small_m_kernel<NG, EPI>), dim3(grid), dim3(NTHREADS), SmallM<NG>::LDS_BYTES + LLADA_SM_TRACE * 8192, stream, q);
    LLADA_CHECK_LAUNCH("gemm_bf16_small_m_kernel");
    return LLADA_OK;
}

template <int NG, int EPI>
int launch_small_m(const GemmParams& p, hipStream_t stream) {
    GemmParams q = p;
    q.no_store = g_no_store;
    q.tiles_m = (int)((p.M + SM_BM - 1) / SM_BM);
    q.tiles_n = (p.N + 32 * NG - 1) / (32 * NG);
    q.n_tiles = 8 * ((q.tiles_n + 7) / 8) * q.tiles_m;   // the block -> tile map deals column tiles to the 8 XCDs
    q.splits = p.plan_splits;
    set_weight_strides(q, p);
    if (q.splits == 1) return launch_small_m_kernel<NG, EPI>(q, q.n_tiles, stream);
    const int rc = launch_small_m_kernel<NG, EPI_PARTIAL>(q, q.n_tiles * q.splits, stream);
    if (rc != LLADA_OK) return rc;
    return finish_split<EPI>(q, stream);
}

template <int EPI>
int dispatch_small_m(const GemmParams& p, hipStream_t stream) {
    switch (p.plan_ng) {
        case 1: return launch_small_m<1, EPI>(p, stream);
        case 2: return launch_small_m<2, EPI>(p, stream);
        case 3: return launch_small_m<3, EPI>(p, stream);
        case 4: return launch_small_m<4, EPI>(p, stream);
        case 6: return launch_small_m<6, EPI>(p, stream);
        default: llada_set_error("gemm: no small-M loop with %d columns", 32 * p.plan_ng); return LLADA_ERR_INVALID;
    }
}

template <int MFR, int EPI>
int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = MFR * 32;
    GemmParams q = p;
    q.band_m = g_band_m;
    q.remap = g_remap;
    q.no_store = g_no_store;
    q.tiles_m = (int)((p.M + BM - 1) / BM);
    q.tiles_n = (p.N + BN - 1) / BN;
    q.n_tiles = q.tiles_m * q.tiles_n;
    const int nk_total = p.K / BK;
    set_weight_strides(q, p);
    q.splits = p.plan_nfh ? p.plan_splits : pick_splits(q.n_tiles, nk_total, p.M, p.N, g_use_pipelined != 0);
    if constexpr (MFR != 8) {     // 128-column tiles: small M only (the 256-row tile never is)
        const int narrow = p.plan_nfh ? (p.plan_nfh == 1 ? p.plan_splits : 0)
                                      : (g_use_pipelined == 1 ? pick_narrow<MFR>(p.M, p.N, nk_total) : 0);
        if (narrow > 0) {
            q.tiles_n = p.N / 128;
            q.n_tiles = q.tiles_m * q.tiles_n;
            q.splits = narrow;
            if (narrow == 1) return launch_main_nt<MFR, EPI, 1>(q, q.n_tiles, stream);
            const int rc = launch_main_nt<MFR, EPI_PARTIAL, 1>(q, q.n_tiles * narrow, stream);
            if (rc != LLADA_OK) return rc;
            return finish_split<EPI>(q, stream);
        }
    }
    if (q.splits == 1) return launch_main_nt<MFR, EPI, 2>(q, q.n_tiles, stream);
    const int rc = launch_main_nt<MFR, EPI_PARTIAL, 2>(q, q.n_tiles * q.splits, stream);
    if (rc != LLADA_OK) return rc;
    return finish_split<EPI>(q, stream);
}

// Pick the M tile (128 / 192 / 256 rows): estimated time = rounds of tiles over the 256 CUs x rows per
// tile x a per-row efficiency factor.  A last round that is only partly filled still takes most of a round
// (a tile needs its full latency; measured ~0.8 of one).  Factors from measurements at K = 4096: the 256-row
// pipelined loop moves 14 % fewer operand bytes per FLOP and runs ~4 % faster per row than the 192-row one
// when its tile count fills the rounds (M = 5120: QKV 353 vs 411 us, gate/up 705 vs 759 us); the 128-row
// loop is 10-17 % slower per row (M = 2304 and 1536, whole rounds: 0.392-0.405 us per row and round against
// 0.345-0.359 for the 192-row tile).
int pick_mfr(int64_t M, int N, int K) {
    (void)K;
    const int cands[3] = {8, 6, 4};
    int best = 8;
    double best_cost = 1e30;
    const int tiles_n = (N + BN - 1) / BN;
    for (int c : cands) {
        const int bm = c * 32;
        const int64_t tiles_m = (M + bm - 1) / bm;
        const int64_t nt = tiles_m * tiles_n;
        const double rounds = (double)(nt / 256) + ((nt % 256) ? 0.8 : 0.0);
        const double cost = rounds * bm * (c == 8 ? 0.96 : (c == 6 ? 1.0 : 1.12));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

template <int EPI>
int dispatch_mfr(const GemmParams& p, int mfr, hipStream_t stream) {
    if (p.plan_ng) return dispatch_small_m<EPI>(p, stream);
    switch (mfr) {
        case 8: return launch<8, EPI>(p, stream);
        case 6: return launch<6, EPI>(p, stream);
        default: return launch<4, EPI>(p, stream);
    }
}

}  // namespace

// Test hook: force the M tile (0 = automatic).
static int g_force_mfr = 0;
extern "C" void llada_gemm_force_tile_rows(int rows) { g_force_mfr = rows / 32; }
extern "C" void llada_gemm_use_pipelined(int on) { g_use_pipelined = on; }
extern "C" void llada_gemm_stream_grid(int n) { g_stream_grid = n > 0 ? (n & ~7) > 0 ? (n & ~7) : 8 : 256; }
extern "C" void llada_gemm_force_splits(int splits) { g_force_splits = splits; }
extern "C" void llada_gemm_force_tile_cols(int cols) { g_force_cols = cols; }
extern "C" void llada_gemm_no_store(int on) { g_no_store = on; }
extern "C" void llada_gemm_force_small_m(int ng) { g_force_small_m = ng; }
static int g_w_tiled = 0;  // timing hook: address W as tile-major whatever the caller says (results are garbage)
extern "C" void llada_gemm_assume_tiled(int on) { g_w_tiled = on; }

extern "C" void llada_gemm_tile_walk(int band_m, int remap) {
    g_band_m = band_m > 0 ? band_m : 8;
    g_remap = remap;
}

// Test hook: the plan of the last main-kernel launch of this thread (what llada_gemm_describe_plan must have predicted).
extern "C" void llada_gemm_last_plan(int32_t* out) {
    for (int i = 0; i < 4; ++i) out[i] = g_last_plan[i];
}

// Test hook (runs on the host only, no GPU needed): the launch plan llada_gemm_bf16 would use for this shape with a
// split-K workspace of `workspace_bytes`: out = {tile rows, tile columns, K ranges, persistent form (0 / 1)}.
extern "C" int llada_gemm_describe_plan(int64_t M, int32_t N, int32_t K, int32_t epilogue, size_t workspace_bytes,
                                        int32_t* out) {
    LLADA_CHECK_ARG(out && M > 0 && N > 0 && K > 0 && K % BK == 0, "gemm_describe_plan: bad argument");
    epilogue &= ~LLADA_GEMM_W_TILED;
    g_slab_bytes = workspace_bytes;
    int mfr = g_force_mfr ? g_force_mfr : pick_mfr(M, N, K);
    int nfh = 2, splits = 0;
    if (!g_force_mfr) {
        const Plan pl = choose_small_m_plan(M, N, K, mfr);
        if (pl.ng) {
            out[0] = SM_BM; out[1] = 32 * pl.ng; out[2] = pl.splits; out[3] = 0;
            return LLADA_OK;
        }
        if (pl.mfr) {
            mfr = pl.mfr;
            nfh = pl.nfh;
            splits = pl.splits;
        }
    }
    const int bm = mfr * 32;
    const int64_t n_tiles = ((M + bm - 1) / bm) * ((N + 128 * nfh - 1) / (128 * nfh));
    if (splits == 0) splits = pick_splits((int)n_tiles, K / BK, M, N, g_use_pipelined != 0);
    const int nk = K / BK / splits;
    const bool nt = g_use_pipelined == 1 && M <= 3 * 192 && mfr != 8;
    const bool stream = mfr != 8 && nfh == 2 && splits == 1 && !nt && nk >= 4 && (nk & 1) == 0 &&
                        (g_use_pipelined == 2 || (g_use_pipelined == 1 && epilogue == LLADA_EPI_SWIGLU && n_tiles >= 2 * g_stream_grid));
    out[0] = bm; out[1] = 128 * nfh; out[2] = splits; out[3] = stream ? 1 : 0;
    return LLADA_OK;
}

// Measurement hook (bench.py): while active, every launch of the GEMM with the chosen epilogue and M is
// bracketed by a pair of HIP events on the launch stream, so the kernel's average duration inside
// the timed region can be read without a profiler.  Not used by the product path.
namespace {
struct GemmProfile {
    int epilogue = -1;
    long long rows = 0;   // only launches with this M are timed
    int capacity = 0, count = 0;
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
} g_prof;
}  // namespace

extern "C" int llada_gemm_profile_begin(int epilogue, int max_samples, long long rows) {
    if (g_prof.start) return LLADA_ERR_INVALID;
    g_prof.start = new hipEvent_t[max_samples];
    g_prof.stop = new hipEvent_t[max_samples];
    for (int i = 0; i < max_samples; ++i) {
        if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess) {
            llada_set_error("gemm_profile_begin: hipEventCreate failed");
            return LLADA_ERR_HIP;
        }
    }
    g_prof.capacity = max_samples;
    g_prof.count = 0;
    g_prof.epilogue = epilogue;
    g_prof.rows = rows;
    return LLADA_OK;
}

// Blocks until the recorded launches have finished; returns their summed duration and count.
extern "C" int llada_gemm_profile_end(double* total_ms, int* n_samples) {
    if (!g_prof.start) return LLADA_ERR_INVALID;
    double tot = 0.0;
    const int n = g_prof.count < g_prof.capacity ? g_prof.count : g_prof.capacity;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_prof.stop[i]);
        (void)hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]);
        tot += ms;
    }
    for (int i = 0; i < g_prof.capacity; ++i) {
        (void)hipEventDestroy(g_prof.start[i]);
        (void)hipEventDestroy(g_prof.stop[i]);
    }
    delete[] g_prof.start;
    delete[] g_prof.stop;
    g_prof = GemmProfile{};
    if (total_ms) *total_ms = tot;
    if (n_samples) *n_samples = n;
    return LLADA_OK;
}

static int gemm_dispatch(GemmParams& p, int epilogue, int mfr, const void* aux, int ldr, const float* rope_cos,
                         const float* rope_sin, int seq_len, int ldc, int N, hipStream_t s);

extern "C" int llada_gemm_bf16(const void* A, const void* W, void* C, int64_t M, int32_t N, int32_t K, int32_t lda,
                               int32_t ldc, int32_t epilogue, const void* aux, int32_t ldr, const float* rope_cos,
                               const float* rope_sin, int32_t seq_len, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return llada_gemm_bf16_positions(A, W, C, M, N, K, lda, ldc, epilogue, aux, ldr, rope_cos, rope_sin, seq_len, nullptr,
                                     workspace, workspace_bytes, stream);
}

int llada_gemm_bf16_positions(const void* A, const void* W, void* C, int64_t M, int32_t N, int32_t K, int32_t lda,
                              int32_t ldc, int32_t epilogue, const void* aux, int32_t ldr, const float* rope_cos,
                              const float* rope_sin, int32_t seq_len, const int32_t* positions, void* workspace,
                              size_t workspace_bytes, void* stream) {
    LLADA_CHECK_ARG(A && W && C, "gemm: null operand");
    LLADA_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    LLADA_CHECK_SUPPORTED(K % BK == 0, "gemm: K=%d must be a multiple of %d", K, BK);
    LLADA_CHECK_SUPPORTED(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
    LLADA_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: A and W must be 16-byte aligned");
    LLADA_CHECK_SUPPORTED(M < (1ll << 31), "gemm: M too large");
    LLADA_CHECK_ARG(lda >= K && (lda % 8) == 0, "gemm: lda=%d must be >= K and a multiple of 8", lda);
    LLADA_CHECK_ARG((ldc % 8) == 0 && ((uintptr_t)C & 15) == 0, "gemm: ldc=%d must be a multiple of 8 and C 16-byte aligned", ldc);
    const bool w_tiled = (epilogue & LLADA_GEMM_W_TILED) != 0;
    epilogue &= ~LLADA_GEMM_W_TILED;
    LLADA_CHECK_SUPPORTED(!w_tiled || N % BN == 0, "gemm: tile-major weights need N=%d to be a multiple of %d", N, BN);
    GemmParams p{};
    p.A = (const bf16_t*)A;
    p.W = (const bf16_t*)W;
    p.C = (bf16_t*)C;
    p.R = (const bf16_t*)aux;
    p.rope_cos = rope_cos;
    p.rope_sin = rope_sin;
    p.M = (int)M;
    p.N = N;
    p.K = K;
    p.lda = lda;
    p.ldc = ldc;
    p.ldr = ldr;
    p.seq_len = seq_len;
    p.positions = positions;
    p.slab = (float*)workspace;
    p.splits = 1;
    if (w_tiled || (g_w_tiled && N % BN == 0)) p.w_row_stride = BK;
    g_slab_bytes = workspace ? workspace_bytes : 0;
    hipStream_t s = (hipStream_t)stream;
    int mfr = g_force_mfr ? g_force_mfr : pick_mfr(M, N, K);
    LLADA_CHECK_ARG(mfr == 8 || mfr == 6 || mfr == 4, "gemm: forced tile rows must be 128, 192 or 256");
    if (!g_force_mfr) {
        const Plan pl = choose_small_m_plan(M, N, K, mfr);
        if (pl.ng) {
            p.plan_ng = pl.ng;
            p.plan_splits = pl.splits;
        } else if (pl.mfr) {
            mfr = pl.mfr;
            p.plan_nfh = pl.nfh;
            p.plan_splits = pl.splits;
        }
    }
    const bool prof = g_prof.epilogue == epilogue && g_prof.rows == M && g_prof.count < g_prof.capacity;
    if (prof) (void)hipEventRecord(g_prof.start[g_prof.count], s);
    const int rc = gemm_dispatch(p, epilogue, mfr, aux, ldr, rope_cos, rope_sin, seq_len, ldc, N, s);
    if (prof) (void)hipEventRecord(g_prof.stop[g_prof.count++], s);
    return rc;
}

static int gemm_dispatch(GemmParams& p, int epilogue, int mfr, const void* aux, int ldr, const float* rope_cos,
                         const float* rope_sin, int seq_len, int ldc, int N, hipStream_t s) {
    switch (epilogue) {
        case LLADA_EPI_PLAIN:
            LLADA_CHECK_ARG(ldc >= N, "gemm: ldc < N");
            return dispatch_mfr<LLADA_EPI_PLAIN>(p, mfr, s);
        case LLADA_EPI_RESIDUAL:
            LLADA_CHECK_ARG(aux != nullptr && ldr >= N && (ldr % 8) == 0 && ((uintptr_t)aux & 15) == 0 && ldc >= N, "gemm: residual needs aux/ldr");
            return dispatch_mfr<LLADA_EPI_RESIDUAL>(p, mfr, s);
        case LLADA_EPI_ROPE_QKV:
            LLADA_CHECK_ARG(rope_cos && rope_sin && seq_len > 0 && ldc >= N, "gemm: rope epilogue needs tables and seq_len");
            LLADA_CHECK_SUPPORTED(N % 3 == 0 && (N / 3) % BN == 0, "gemm: rope epilogue needs d_model %% 256 == 0");
            p.rope_cols = 2 * (N / 3);
            return dispatch_mfr<LLADA_EPI_ROPE_QKV>(p, mfr, s);
        case LLADA_EPI_SWIGLU:
            LLADA_CHECK_SUPPORTED(N % 64 == 0, "gemm: swiglu epilogue needs N %% 64 == 0");
            LLADA_CHECK_ARG(ldc >= N / 2, "gemm: ldc < N/2");
            return dispatch_mfr<LLADA_EPI_SWIGLU>(p, mfr, s);
        default:
            llada_set_error("gemm: unknown epilogue %d", epilogue);
            return LLADA_ERR_INVALID;
    }
}

// X[M, N] += A[M, K] W[N, K]^T and H = RMSNorm(X) * norm_w: a residual projection followed by the norm
// that consumes it.  When the GEMM runs split-K (small M) the second pass produces both outputs; otherwise
// this is llada_gemm_bf16(EPI_RESIDUAL) followed by llada_rmsnorm_bf16.  Bit-identical either way.
extern "C" int llada_gemm_residual_rmsnorm_bf16(const void* A, const void* W, void* X, void* H, const void* norm_w,
                                                float eps, int64_t M, int32_t N, int32_t K, int32_t lda,
                                                int32_t w_tiled, void* workspace, size_t workspace_bytes,
                                                void* stream) {
    LLADA_CHECK_ARG(A && W && X && H && norm_w, "gemm_residual_rmsnorm: null operand");
    LLADA_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_residual_rmsnorm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    LLADA_CHECK_SUPPORTED(K % BK == 0, "gemm: K=%d must be a multiple of %d", K, BK);
    LLADA_CHECK_SUPPORTED(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
    LLADA_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)X & 15) == 0 &&
                        ((uintptr_t)H & 15) == 0 && ((uintptr_t)norm_w & 15) == 0,
                    "gemm_residual_rmsnorm: operands must be 16-byte aligned");
    LLADA_CHECK_SUPPORTED(M < (1ll << 31), "gemm: M too large");
    LLADA_CHECK_ARG(lda >= K && (lda % 8) == 0, "gemm: lda=%d must be >= K and a multiple of 8", lda);
    LLADA_CHECK_SUPPORTED(!w_tiled || N % BN == 0, "gemm: tile-major weights need N=%d to be a multiple of %d", N, BN);
    GemmParams p{};
    p.A = (const bf16_t*)A;
    p.W = (const bf16_t*)W;
    p.C = (bf16_t*)X;
    p.R = (const bf16_t*)X;
    p.M = (int)M;
    p.N = N;
    p.K = K;
    p.lda = lda;
    p.ldc = N;
    p.ldr = N;
    p.slab = (float*)workspace;
    p.splits = 1;
    int fused = 0;
    p.norm_w = (const bf16_t*)norm_w;
    p.norm_out = (bf16_t*)H;
    p.norm_eps = eps;
    p.norm_fused = &fused;
    if (w_tiled || (g_w_tiled && N % BN == 0)) p.w_row_stride = BK;
    g_slab_bytes = workspace ? workspace_bytes : 0;
    hipStream_t s = (hipStream_t)stream;
    int mfr = g_force_mfr ? g_force_mfr : pick_mfr(M, N, K);
    LLADA_CHECK_ARG(mfr == 8 || mfr == 6 || mfr == 4, "gemm: forced tile rows must be 128, 192 or 256");
    if (!g_force_mfr) {
        const Plan pl = choose_small_m_plan(M, N, K, mfr);
        if (pl.ng) {
            p.plan_ng = pl.ng;
            p.plan_splits = pl.splits;
        } else if (pl.mfr) {
            mfr = pl.mfr;
            p.plan_nfh = pl.nfh;
            p.plan_splits = pl.splits;
        }
    }
    const int rc = dispatch_mfr<LLADA_EPI_RESIDUAL>(p, mfr, s);
    if (rc != LLADA_OK || fused) return rc;
    return llada_rmsnorm_bf16(X, norm_w, H, M, N, eps, nullptr, stream);
}

// Bytes of float32 scratch with which llada_gemm_bf16 can balance this shape over all CUs (0 if it
// needs none).  Any smaller workspace (or none) is valid: the call then falls back to whole tiles
// per workgroup.
extern "C" size_t llada_gemm_workspace_bytes(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return 0;
    const int mfr = g_force_mfr ? g_force_mfr : pick_mfr(M, N, K);
    const int bm = mfr * 32;
    const long long nt = ((M + bm - 1) / bm) * ((N + BN - 1) / BN);
    return nt >= 192 ? 0 : (size_t)16 * M * N * 4;
}
