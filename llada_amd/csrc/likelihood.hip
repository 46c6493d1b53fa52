// Monte-Carlo masked log-likelihood pieces around the forward (get_log_likelihood.py:7-26, 66-75).
//
//   ll_mask_kernel   : forward_process -- per row, counts[r] answer positions become the mask token,
//                      chosen by an injected permutation (parity tests) or by the counts[r] smallest
//                      Philox keys of the row (uniformly random subset, like masking a randperm prefix).
//   masked_ce_kernel : one workgroup per answer-position logits row; rows whose noisy token is not the
//                      mask contribute 0.  Fuses the CFG combine (get_log_likelihood.py:41-42),
//                      F.cross_entropy(reduction='none') on bf16 logits (:72) and the division by
//                      p_mask; a second tiny kernel adds the row losses in a fixed order (float64).
#include "common.h"
#include "philox.h"
#include "rowregs.h"
#include "rowstream.h"

namespace {

__global__ void __launch_bounds__(256) ll_mask_kernel(const int64_t* __restrict__ seq, int64_t* __restrict__ noisy,
                                                      int l, int prompt_len, int64_t mask_id,
                                                      const int32_t* __restrict__ counts,
                                                      const int32_t* __restrict__ perms, uint64_t seed,
                                                      uint32_t iteration, uint32_t row_offset) {
    extern __shared__ uint32_t keys[];  // [target_len] when perms == nullptr
    const int r = blockIdx.x, tid = threadIdx.x;
    const int T = l - prompt_len;
    const int count = counts[r];
    for (int i = tid; i < prompt_len; i += 256) noisy[(size_t)r * l + i] = seq[(size_t)r * l + i];
    if (perms) {
        for (int j = tid; j < T; j += 256) {
            const size_t p = (size_t)r * l + prompt_len + j;
            noisy[p] = perms[(size_t)r * T + j] < count ? mask_id : seq[p];
        }
        return;
    }
    uint32_t k0, k1;
    philox_key(seed, LLADA_STREAM_LL, k0, k1);
    for (int j = tid; j < T; j += 256)
        keys[j] = philox4x32_10((uint32_t)j, 0u, row_offset + (uint32_t)r, iteration, k0, k1).v[0];
    __syncthreads();
    for (int j = tid; j < T; j += 256) {
        const uint32_t kj = keys[j];
        int rank = 0;
        for (int i = 0; i < T; ++i) {
            const uint32_t ki = keys[i];
            rank += (ki < kj) || (ki == kj && i < j);
        }
        const size_t p = (size_t)r * l + prompt_len + j;
        noisy[p] = rank < count ? mask_id : seq[p];
    }
}

// Ordered compaction of the masked answer positions: row r writes the flat indices r * l + column of its
// mask tokens, ascending, starting at sum(counts[0 .. r)) -- the host knows every count (it drew them), so
// the forward can be asked for exactly these logits rows without reading anything back.
__global__ void __launch_bounds__(256) ll_masked_rows_kernel(const int64_t* __restrict__ noisy, int l, int prompt_len,
                                                             int64_t mask_id, const int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ rows_out) {
    __shared__ int red[4];
    __shared__ int wave_tot[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int off = 0;
    for (int i = tid; i < r; i += 256) off += counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o);
    if (lane == 0) red[wave] = off;
    __syncthreads();
    int base = red[0] + red[1] + red[2] + red[3];
    const int limit = base + counts[r];          // never write past this row's share
    const int T = l - prompt_len;
    for (int j0 = 0; j0 < T; j0 += 256) {
        const int j = j0 + tid;
        const bool flag = j < T && noisy[(size_t)r * l + prompt_len + j] == mask_id;
        const unsigned long long bal = __ballot(flag);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();                         // wave_tot of the previous trip has been read
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int wbase = base;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        if (flag && wbase + before < limit) rows_out[wbase + before] = r * l + prompt_len + j;
        base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    }
}

// One 1024-thread workgroup per row; the row is read from memory once and kept on chip (rowregs.h).
__global__ void __launch_bounds__(ROW_THREADS) masked_ce_kernel(const bf16_t* __restrict__ logits,
                                                                const bf16_t* __restrict__ uncond, float cfg1,
                                                                const int32_t* __restrict__ row_pos, int V,
                                                                const int64_t* __restrict__ noisy,
                                                                const int64_t* __restrict__ seq, int l, int64_t mask_id,
                                                                const float* __restrict__ p_mask_rows, int round_ce,
                                                                float* __restrict__ row_loss) {
    __shared__ float red[2][ROW_THREADS / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char row_lds[];   // ROW_LDS_BYTES
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = row_pos ? row_pos[r] : r;
    if (noisy[pos] != mask_id) {
        if (tid == 0) row_loss[r] = 0.f;
        return;
    }
    const bf16_t* lc = logits + (size_t)r * V;
    const bf16_t* lu = uncond ? uncond + (size_t)r * V : nullptr;
    RowRegs row;
    row.load(lc, lu, cfg1, V / 8, tid, row_lds);
    const float m = block_max16(row.max(), red[0], lane, wave);
    const float tot = block_sum16(row.sum_exp(m), red[1], lane, wave);
    if (tid == 0) {
        const int64_t tgt = seq[pos];
        if (tgt < 0 || tgt >= V) {       // F.cross_entropy raises for such a target; here the sum turns NaN (no read outside the row)
            row_loss[r] = NAN;
            return;
        }
        float lt = bf16_to_f32(lc[tgt]);
        if (lu) lt = cfg_combine(lt, bf16_to_f32(lu[tgt]), cfg1);
        float logp = (lt - m) - logf(tot);
        if (round_ce) logp = round_bf16(logp);
        row_loss[r] = -logp / p_mask_rows[pos / l];
    }
}

// Streamed variant (rowstream.h): block b = piece b % splits of logits row b / splits.
__global__ void __launch_bounds__(RS_THREADS) ce_stream_kernel(const bf16_t* __restrict__ logits,
                                                               const bf16_t* __restrict__ uncond, float cfg1,
                                                               const int32_t* __restrict__ row_pos, int V,
                                                               const int64_t* __restrict__ noisy, int64_t mask_id,
                                                               int splits, int cps, RowPart* __restrict__ parts) {
    __shared__ float red_f[2 * RS_THREADS / 64];
    __shared__ int red_i[RS_THREADS / 64];
    const int r = blockIdx.x / splits, sp = blockIdx.x - r * splits;
    const int pos = row_pos ? row_pos[r] : r;
    if (noisy[pos] != mask_id) return;
    const int chunks = V / 8;
    const int c_begin = sp * cps, c_end = min(chunks, c_begin + cps);
    const bf16_t* lc = logits + (size_t)r * V;
    const bf16_t* lu = uncond ? uncond + (size_t)r * V : nullptr;
    const RowPart p = row_stream_part<true, false>(lc, lu, cfg1, -1, -1, c_begin, c_end, red_f, red_i);
    if (threadIdx.x == 0) parts[(size_t)r * splits + sp] = p;
}

// One workgroup: merges the pieces of every row, forms CE / p_mask per row and adds the rows in the
// same fixed order as sum_rows_kernel.
__global__ void __launch_bounds__(256) ce_merge_kernel(const bf16_t* __restrict__ logits, const bf16_t* __restrict__ uncond,
                                                       float cfg1, const int32_t* __restrict__ row_pos, int V,
                                                       const int64_t* __restrict__ noisy, const int64_t* __restrict__ seq,
                                                       int l, int64_t mask_id, const float* __restrict__ p_mask_rows,
                                                       int round_ce, int splits, const RowPart* __restrict__ parts,
                                                       int n_rows, float* __restrict__ row_loss, double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int r = threadIdx.x; r < n_rows; r += 256) {
        const int pos = row_pos ? row_pos[r] : r;
        float loss = 0.f;
        if (noisy[pos] == mask_id) {
            const RowPart* p = parts + (size_t)r * splits;
            float M = p[0].m, S = p[0].s;
            for (int k = 1; k < splits; ++k) rs_merge(M, S, p[k].m, p[k].s);
            const int64_t tgt = seq[pos];
            if (tgt < 0 || tgt >= V) {   // F.cross_entropy raises for such a target; here the sum turns NaN
                loss = NAN;
            } else {
                float lt = bf16_to_f32(logits[(size_t)r * V + tgt]);
                if (uncond) lt = cfg_combine(lt, bf16_to_f32(uncond[(size_t)r * V + tgt]), cfg1);
                float logp = (lt - M) - logf(S);
                if (round_ce) logp = round_bf16(logp);
                loss = -logp / p_mask_rows[pos / l];
            }
        }
        row_loss[r] = loss;
        acc += (double)loss;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

__global__ void __launch_bounds__(256) sum_rows_kernel(const float* __restrict__ row_loss, int n,
                                                       double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)row_loss[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

}  // namespace

extern "C" int llada_ll_mask(const int64_t* seq, int64_t* noisy, int32_t b, int32_t l, int32_t prompt_len,
                             int64_t mask_id, const int32_t* counts, const int32_t* perms, uint64_t seed,
                             uint32_t iteration, uint32_t row_offset, void* stream) {
    LLADA_CHECK_ARG(seq && noisy && counts && b > 0 && l > 0, "ll_mask: bad argument");
    LLADA_CHECK_ARG(prompt_len >= 0 && prompt_len < l, "ll_mask: prompt_len must leave at least one answer token");
    const int T = l - prompt_len;
    LLADA_CHECK_SUPPORTED(T <= 15360, "ll_mask: answer too long (%d)", T);
    hipLaunchKernelGGL(ll_mask_kernel, dim3(b), dim3(256), perms ? 0 : (size_t)T * sizeof(uint32_t),
                       (hipStream_t)stream, seq, noisy, l, prompt_len, mask_id, counts, perms, seed, iteration,
                       row_offset);
    LLADA_CHECK_LAUNCH("ll_mask_kernel");
    return LLADA_OK;
}

extern "C" int llada_ll_masked_rows(const int64_t* noisy, int32_t b, int32_t l, int32_t prompt_len, int64_t mask_id,
                                    const int32_t* counts, int32_t* rows_out, void* stream) {
    LLADA_CHECK_ARG(noisy && counts && rows_out && b > 0 && l > 0, "ll_masked_rows: bad argument");
    LLADA_CHECK_ARG(prompt_len >= 0 && prompt_len < l, "ll_masked_rows: prompt_len must leave at least one answer token");
    LLADA_CHECK_SUPPORTED((int64_t)b * l < 0x7fffffffLL, "ll_masked_rows: b*l too large");
    hipLaunchKernelGGL(ll_masked_rows_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, noisy, l, prompt_len, mask_id,
                       counts, rows_out);
    LLADA_CHECK_LAUNCH("ll_masked_rows_kernel");
    return LLADA_OK;
}

extern "C" int llada_masked_ce(const void* logits, const void* uncond_logits, double cfg_scale, const int32_t* row_pos,
                               int32_t n_rows, int32_t vocab, const int64_t* noisy, const int64_t* seq, int32_t b,
                               int32_t l, int64_t mask_id, const float* p_mask_rows, int32_t round_ce_bf16,
                               float* row_loss, double* loss_sum, void* workspace, size_t workspace_bytes,
                               void* stream) {
    LLADA_CHECK_ARG(logits && noisy && seq && p_mask_rows && row_loss && loss_sum, "masked_ce: null tensor");
    LLADA_CHECK_ARG(n_rows > 0 && vocab > 0 && b > 0 && l > 0, "masked_ce: bad shape");
    LLADA_CHECK_SUPPORTED(vocab % 8 == 0, "masked_ce: vocab=%d must be a multiple of 8", vocab);
    LLADA_CHECK_ARG(row_pos != nullptr || n_rows == b * l, "masked_ce: row_pos is required unless n_rows == b*l");
    hipStream_t s = (hipStream_t)stream;
    if (workspace != nullptr) {
        LLADA_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "masked_ce: workspace must be 16-byte aligned");
        const int chunks = vocab / 8;
        const int splits = rs_pick_splits(n_rows, chunks, llada_rows_forced_splits);
        if ((size_t)n_rows * splits * sizeof(RowPart) > workspace_bytes) {
            llada_set_error("masked_ce: workspace of %zu bytes, %zu needed (llada_rows_workspace_bytes)", workspace_bytes,
                            (size_t)n_rows * splits * sizeof(RowPart));
            return LLADA_ERR_WORKSPACE;
        }
        LLADA_CHECK_SUPPORTED((int64_t)n_rows * splits < 0x7fffffffLL, "masked_ce: grid too large");
        const int cps = (chunks + splits - 1) / splits;
        const float cfg1 = (float)(cfg_scale + 1.0);
        hipLaunchKernelGGL(ce_stream_kernel, dim3(n_rows * splits), dim3(RS_THREADS), 0, s, (const bf16_t*)logits,
                           (const bf16_t*)uncond_logits, cfg1, row_pos, vocab, noisy, mask_id, splits, cps,
                           (RowPart*)workspace);
        LLADA_CHECK_LAUNCH("ce_stream_kernel");
        hipLaunchKernelGGL(ce_merge_kernel, dim3(1), dim3(256), 0, s, (const bf16_t*)logits, (const bf16_t*)uncond_logits,
                           cfg1, row_pos, vocab, noisy, seq, l, mask_id, p_mask_rows, round_ce_bf16, splits,
                           (const RowPart*)workspace, n_rows, row_loss, loss_sum);
        LLADA_CHECK_LAUNCH("ce_merge_kernel");
        return LLADA_OK;
    }
    LLADA_CHECK_SUPPORTED(vocab <= ROW_MAX_VOCAB, "masked_ce: vocab=%d exceeds %d (one row must fit on chip)", vocab, ROW_MAX_VOCAB);
    static OncePerDevice attr_set;
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)masked_ce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           ROW_LDS_BYTES);
        if (e != hipSuccess) {
            llada_set_error("hipFuncSetAttribute(masked_ce_kernel): %s", hipGetErrorString(e));
            return LLADA_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(masked_ce_kernel, dim3(n_rows), dim3(ROW_THREADS), ROW_LDS_BYTES, s, (const bf16_t*)logits,
                       (const bf16_t*)uncond_logits, (float)(cfg_scale + 1.0), row_pos, vocab, noisy, seq, l, mask_id,
                       p_mask_rows, round_ce_bf16, row_loss);
    LLADA_CHECK_LAUNCH("masked_ce_kernel");
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, s, row_loss, n_rows, loss_sum);
    LLADA_CHECK_LAUNCH("sum_rows_kernel");
    return LLADA_OK;
}
