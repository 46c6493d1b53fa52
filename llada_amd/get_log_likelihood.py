"""Drop-in for /root/reference/get_log_likelihood.py::get_log_likelihood -- same positional
arguments and defaults (:46-47), returns a Python float.

Per Monte-Carlo iteration: forward_process (:7-26) runs as one kernel (stratified mask counts are
host arithmetic, the random subset comes from the library's Philox stream or from injected
permutations), the forward computes logits for the masked answer positions only (the reference
indexes `logits[mask_index]`, :72; the positions are compacted on the device and their number is
host arithmetic), and one kernel fuses the CFG combine, the cross-entropy on the masked
positions and the division by p_mask (:72-73).  The host synchronises once, after the last
iteration (the reference calls .item() every iteration, :75).
"""
import numpy as np
import torch

from . import _lib, rng
from .model import LLaDAModel, check_token_ids
from .parallel import combine_loss_sums, shard_bounds


FUSED_FORWARD_MAX_ROWS = 64     # sequences per fused forward (get_log_likelihood(fuse_iterations=True))


def stratified_counts(k, b, target_len):
    """get_log_likelihood.py:13-14: round(linspace(k, k+(b-1)*(T/b), b)) in float32 (torch.linspace
    evaluates start + i*step in the first half and end - (b-1-i)*step in the second),
    half-to-even, then ((x-1) mod T) + 1."""
    start = np.float32(float(k))
    end = np.float32(k + (b - 1) * (target_len / b))
    if b == 1:
        vals = np.array([start], dtype=np.float32)
    else:
        step = np.float32((end - start) / np.float32(b - 1))
        i = np.arange(b)
        lo = (start + step * i.astype(np.float32)).astype(np.float32)
        hi = (end - step * (b - 1 - i).astype(np.float32)).astype(np.float32)
        vals = np.where(i < b // 2, lo, hi).astype(np.float32)
    x = np.rint(vals).astype(np.int64)
    x = ((x - 1) % target_len) + 1
    assert x.min() >= 1 and x.max() <= target_len                                   # :15
    return x


def forward_process(batch, prompt_index, mask_id, *, k=None, perms=None, seed=0, iteration=0, row_offset=0):
    """get_log_likelihood.py:7-26 on the device.  Returns (noisy_batch, p_mask [b, l])."""
    lib = _lib.load()
    b, l = batch.shape
    prompt_len = int(prompt_index.sum().item()) if torch.is_tensor(prompt_index) else int(prompt_index)
    target_len = l - prompt_len
    if k is None:
        k = rng.ll_k(seed, iteration, target_len)
    counts = stratified_counts(int(k), b, target_len)
    counts_t = torch.from_numpy(counts.astype(np.int32)).to(batch.device)
    perms_t = None if perms is None else perms.to(device=batch.device, dtype=torch.int32).contiguous()
    noisy = torch.empty_like(batch)
    _lib.check(lib.llada_ll_mask(batch.contiguous().data_ptr(), noisy.data_ptr(), b, l, prompt_len, mask_id,
                                 counts_t.data_ptr(), _lib.ptr(perms_t), seed, iteration, row_offset,
                                 _lib.stream_ptr()))
    p = torch.from_numpy((counts.astype(np.float32) / np.float32(target_len))).to(batch.device)
    return noisy, p[:, None].repeat(1, l)


@torch.no_grad()
def get_log_likelihood(model, prompt, answer, mc_num=128, batch_size=16, cfg_scale=0., mask_id=126336, *,
                       seed=0, k_fn=None, perm_fn=None, round_ce_bf16=True, rank=0, world_size=1,
                       process_group=None, return_loss_sums=False, fuse_iterations=True):
    '''
    Args (get_log_likelihood.py:48-59):
        model: Mask predictor.  prompt: tensor (l1).  answer: tensor (l2).
        mc_num: Monte Carlo estimation times.  batch_size: Mini batch size.
        cfg_scale: Unsupervised classifier-free guidance scale.  mask_id: 126336.
    Keyword-only extensions:
        seed: key of the library's Philox mask stream.
        k_fn(target_len) / perm_fn(target_len): replace torch.randint / torch.randperm (:11, :20).
        round_ce_bf16: round each token's cross-entropy to bf16 as F.cross_entropy does on bf16
            logits (True reproduces the reference; False keeps float32).
        rank / world_size / process_group: data-parallel split of every iteration's batch_size
            rows over ranks; the per-iteration loss sums are all-reduced once at the end.
        return_loss_sums: return this rank's per-iteration sums of CE / p_mask (float64 device tensor
            [mc_num // batch_size]) instead of reducing them: the rank's contribution before the all-reduce.
        fuse_iterations: when this rank's rows of ALL iterations together are at most FUSED_FORWARD_MAX_ROWS
            sequences (the batch split over many GPUs: 4 iterations x 4 rows at mc_num 128, batch_size 32 on 8
            GPUs), run them through one forward instead of one per iteration (:66-70 loop; the rows are
            independent).  Same masks and sums; False keeps one forward per iteration.
    '''
    lib = _lib.load()
    device = torch.device(model.device)
    seq = torch.concatenate([prompt.to(device), answer.to(device)])[None, :].to(torch.long)     # :61
    seq = seq.repeat((batch_size, 1)).contiguous()                                              # :62
    l1, l = len(prompt), seq.shape[1]
    T = l - l1
    assert T >= 1
    n_iter = mc_num // batch_size                                                               # :66
    if n_iter == 0:
        raise ZeroDivisionError("division by zero")                                             # :77 with no iteration
    fast = isinstance(model, LLaDAModel)
    if fast:
        check_token_ids(seq[0], model.cfg.vocab_size, mask_id, "prompt + answer")   # F.cross_entropy / the embedding would raise
        # A literal mask token in the prompt or the answer is part of `mask_index` in the reference
        # (get_log_likelihood.py:68 compares the whole row), so it adds a CE / p_mask term there.  The compacted
        # fast path below only looks at the answer positions the masking itself chose: take the full-logits
        # path (every position of every row, as the reference) in that rare case.  One host read per call.
        if bool((seq[0] == mask_id).any().item()):
            fast = False

    # this rank's contiguous share of the batch rows
    bounds = shard_bounds(batch_size, world_size)
    r0, r1 = bounds[rank], bounds[rank + 1]
    b_loc = r1 - r0

    counts_all = np.zeros((n_iter, batch_size), dtype=np.int32)
    perms_all = None
    for it in range(n_iter):
        k = int(k_fn(T)) if k_fn is not None else rng.ll_k(seed, it, T)
        counts_all[it] = stratified_counts(k, batch_size, T)
        if perm_fn is not None:
            if perms_all is None:
                perms_all = np.zeros((n_iter, batch_size, T), dtype=np.int32)
            for r in range(batch_size):
                perms_all[it, r] = np.asarray(perm_fn(T))
    loss_sums = torch.zeros(n_iter, dtype=torch.float64, device=device)
    if b_loc > 0:
        counts_t = torch.from_numpy(np.ascontiguousarray(counts_all[:, r0:r1])).to(device)
        pmask_t = torch.from_numpy(np.ascontiguousarray(
            counts_all[:, r0:r1].astype(np.float32) / np.float32(T))).to(device)           # :26
        perms_t = None if perms_all is None else torch.from_numpy(np.ascontiguousarray(perms_all[:, r0:r1])).to(device)
        seq_loc = seq[:b_loc]
        noisy = torch.empty_like(seq_loc)
        # answer positions of this rank's rows (protocol-only models return every row anyway)
        n_ans = b_loc * T
        masked_per_iter = counts_all[:, r0:r1].astype(np.int64).sum(axis=1)        # known on the host
        rows_buf = torch.empty(n_ans, dtype=torch.int32, device=device)
        row_loss = torch.empty(n_ans if fast else b_loc * l, dtype=torch.float32, device=device)
        ws = torch.empty(lib.llada_rows_workspace_bytes(row_loss.numel()), dtype=torch.uint8, device=device)
        stream = _lib.stream_ptr()
        # Few rows per rank (the batch split over many GPUs): all iterations of this rank share ONE forward.  Every
        # iteration's masks, masked positions, p_mask and loss sum are exactly those of the loop below; only the
        # number of rows per forward differs (other GEMM tilings, i.e. the last bits of the bf16 roundings).
        n_seq_all = n_iter * b_loc * (2 if cfg_scale > 0. else 1)
        if fast and fuse_iterations and n_iter > 1 and n_seq_all <= FUSED_FORWARD_MAX_ROWS:
            noisy_all = torch.empty((n_iter * b_loc, l), dtype=torch.long, device=device)
            rows_all = torch.empty(int(masked_per_iter.sum()), dtype=torch.int32, device=device)
            offs = np.concatenate([[0], np.cumsum(masked_per_iter)]).astype(np.int64)
            for it in range(n_iter):
                noisy_it = noisy_all[it * b_loc:(it + 1) * b_loc]
                _lib.check(lib.llada_ll_mask(seq_loc.data_ptr(), noisy_it.data_ptr(), b_loc, l, l1, mask_id,
                                             counts_t[it].data_ptr(), None if perms_t is None else perms_t[it].data_ptr(),
                                             seed, it, r0, stream))                             # :67
                _lib.check(lib.llada_ll_masked_rows(noisy_it.data_ptr(), b_loc, l, l1, mask_id, counts_t[it].data_ptr(),
                                                    rows_all[int(offs[it]):].data_ptr(), stream))
            # token indices inside the iteration's own rows (the cross-entropy kernel's view) and inside the fused batch
            rows_fwd = rows_all + torch.repeat_interleave(
                torch.arange(n_iter, device=device, dtype=torch.int32) * (b_loc * l),
                torch.from_numpy(masked_per_iter).to(device))
            n_all = rows_all.numel()
            if cfg_scale > 0.:                                                                  # :30-35
                un_all = noisy_all.clone()
                un_all[:, :l1] = mask_id
                batch = torch.cat([noisy_all, un_all])
                rows_fwd = torch.cat([rows_fwd, rows_fwd + n_iter * b_loc * l])
            else:
                batch = noisy_all
            lg = model.forward_rows(batch, None, rows_fwd.contiguous())                         # :38, once
            for it in range(n_iter):
                o0, n_rows = int(offs[it]), int(masked_per_iter[it])
                cond = lg[o0:o0 + n_rows]
                un_ptr = lg[n_all + o0:n_all + o0 + n_rows].data_ptr() if cfg_scale > 0. else None
                _lib.check(lib.llada_masked_ce(cond.data_ptr(), un_ptr, float(cfg_scale), rows_all[o0:].data_ptr(), n_rows,
                                               cond.shape[-1], noisy_all[it * b_loc:].data_ptr(), seq_loc.data_ptr(), b_loc, l,
                                               mask_id, pmask_t[it].data_ptr(), 1 if round_ce_bf16 else 0,
                                               row_loss.data_ptr(), loss_sums[it:].data_ptr(), ws.data_ptr(), ws.numel(),
                                               stream))                                         # :72
            n_iter_loop = 0
        else:
            n_iter_loop = n_iter
        for it in range(n_iter_loop):
            _lib.check(lib.llada_ll_mask(seq_loc.data_ptr(), noisy.data_ptr(), b_loc, l, l1, mask_id,
                                         counts_t[it].data_ptr(), None if perms_t is None else perms_t[it].data_ptr(),
                                         seed, it, r0, stream))                                 # :67
            un_ptr = None
            if cfg_scale > 0.:                                                                  # :30-35
                un_batch = noisy.clone()
                un_batch[:, :l1] = mask_id
                batch = torch.cat([noisy, un_batch])
            else:
                batch = noisy
            if fast:
                # logits only where :72 reads them (`logits[mask_index]`): the masked answer positions,
                # compacted on the device; their number is the sum of this iteration's counts
                n_rows = int(masked_per_iter[it])
                rows = rows_buf[:n_rows]
                _lib.check(lib.llada_ll_masked_rows(noisy.data_ptr(), b_loc, l, l1, mask_id, counts_t[it].data_ptr(),
                                                    rows.data_ptr(), stream))
                rows2 = torch.cat([rows, rows + b_loc * l]).contiguous() if cfg_scale > 0. else rows
                lg = model.forward_rows(batch, None, rows2)                                     # :38
                cond = lg[:n_rows]
                if cfg_scale > 0.:
                    un_ptr = lg[n_rows:].data_ptr()
                row_pos, nr = rows, n_rows
            else:
                lg = model(batch).logits.to(device=device, dtype=torch.bfloat16).contiguous()
                cond = lg[:b_loc].reshape(b_loc * l, -1)
                if cfg_scale > 0.:
                    un_ptr = lg[b_loc:].reshape(b_loc * l, -1).data_ptr()
                row_pos, nr = None, b_loc * l
            _lib.check(lib.llada_masked_ce(cond.data_ptr(), un_ptr, float(cfg_scale), _lib.ptr(row_pos), nr,
                                           cond.shape[-1], noisy.data_ptr(), seq_loc.data_ptr(), b_loc, l, mask_id,
                                           pmask_t[it].data_ptr(), 1 if round_ce_bf16 else 0, row_loss.data_ptr(),
                                           loss_sums[it:].data_ptr(), ws.data_ptr(), ws.numel(), stream))  # :72
    if return_loss_sums:
        return loss_sums
    return combine_loss_sums(loss_sums, batch_size, world_size, process_group)


@torch.no_grad()
def get_log_likelihood_many(model, pairs, mc_num=128, batch_size=16, cfg_scale=0., mask_id=126336, *, seeds=None,
                            max_rows=64, round_ce_bf16=True):
    '''The estimator of get_log_likelihood (get_log_likelihood.py:46-77) for several (prompt, answer) pairs at
    once: the Monte-Carlo rows of different requests share forwards (the reference's harness loops over requests,
    eval_llada.py:231-238, so a request with mc_num = batch_size = 1 runs the model on a single short row).
    Rows are right-padded to the longest of their forward and the padding is masked as attention keys, which
    leaves every real position's arithmetic unchanged; ``seeds[i]`` keys request i's masks exactly as
    ``get_log_likelihood(..., seed=seeds[i])`` does.  Padding and masking change nothing (a right-padded, masked
    row gives bit-identical logits to the bare row); what differs from the one-by-one calls is the number of rows
    per forward, hence the GEMM tiling and the last bits of every bf16 rounding -- the same run-to-run noise the
    reference has between batch sizes.  ``max_rows``: rows per forward.  llada_amd.LLaDAModel only.
    Returns a list of floats.'''
    assert isinstance(model, LLaDAModel), "get_log_likelihood_many needs an llada_amd.LLaDAModel"
    lib = _lib.load()
    device = torch.device(model.device)
    n_iter = mc_num // batch_size                                                               # :66
    if n_iter == 0:
        raise ZeroDivisionError("division by zero")
    if seeds is None:
        seeds = list(range(len(pairs)))
    assert len(seeds) == len(pairs)
    # ids outside the vocabulary raise as in the reference; a request that contains a literal mask token needs
    # the full-logits path of get_log_likelihood (get_log_likelihood.py:68 counts it) and is run on its own
    special = []
    for i, (prompt, answer) in enumerate(pairs):
        both = torch.cat([prompt.reshape(-1), answer.reshape(-1)]).to(device)
        check_token_ids(both, model.cfg.vocab_size, mask_id, f"request {i}")
        if bool((both == mask_id).any().item()):
            special.append(i)
    if special:
        keep = [i for i in range(len(pairs)) if i not in special]
        out = [None] * len(pairs)
        if keep:
            rest = get_log_likelihood_many(model, [pairs[i] for i in keep], mc_num, batch_size, cfg_scale, mask_id,
                                           seeds=[seeds[i] for i in keep], max_rows=max_rows, round_ce_bf16=round_ce_bf16)
            for i, v in zip(keep, rest):
                out[i] = v
        for i in special:
            out[i] = get_log_likelihood(model, pairs[i][0], pairs[i][1], mc_num, batch_size, cfg_scale, mask_id,
                                        seed=seeds[i], round_ce_bf16=round_ce_bf16)
        return out
    stream = _lib.stream_ptr()
    # one group = the batch_size rows of one iteration of one request
    groups = []
    for i, (prompt, answer) in enumerate(pairs):
        seq = torch.cat([prompt.to(device), answer.to(device)]).to(torch.long)                  # :61
        l1, l = len(prompt), seq.numel()
        T = l - l1
        assert T >= 1
        for it in range(n_iter):
            counts = stratified_counts(rng.ll_k(seeds[i], it, T), batch_size, T)                # :10-14
            groups.append((i, it, seq, l1, l, counts))
    sums = np.zeros((len(pairs), n_iter), dtype=np.float64)
    g0 = 0
    while g0 < len(groups):
        g1 = g0 + 1
        while g1 < len(groups) and (g1 + 1 - g0) * batch_size <= max(max_rows, batch_size):
            g1 += 1
        chunk = groups[g0:g1]
        R, Lmax = len(chunk) * batch_size, max(g[4] for g in chunk)
        noisy = torch.zeros((R, Lmax), dtype=torch.long, device=device)
        seqs = torch.zeros((R, Lmax), dtype=torch.long, device=device)
        keym = torch.zeros((R, Lmax), dtype=torch.long, device=device)
        answer_region = torch.zeros((R, Lmax), dtype=torch.bool, device=device)
        prompt_region = torch.zeros((R, Lmax), dtype=torch.bool, device=device)
        p_mask = np.zeros(R, dtype=np.float32)
        for j, (i, it, seq, l1, l, counts) in enumerate(chunk):
            r0 = j * batch_size
            rep = seq[None, :].repeat(batch_size, 1).contiguous()                               # :62
            out = torch.empty_like(rep)
            counts_t = torch.from_numpy(counts.astype(np.int32)).to(device)
            _lib.check(lib.llada_ll_mask(rep.data_ptr(), out.data_ptr(), batch_size, l, l1, mask_id, counts_t.data_ptr(),
                                         None, seeds[i], it, 0, stream))                        # :67
            noisy[r0:r0 + batch_size, :l] = out
            seqs[r0:r0 + batch_size, :l] = rep
            keym[r0:r0 + batch_size, :l] = 1
            answer_region[r0:r0 + batch_size, l1:l] = True
            prompt_region[r0:r0 + batch_size, :l1] = True
            p_mask[r0:r0 + batch_size] = counts.astype(np.float32) / np.float32(l - l1)         # :26
        rows = torch.nonzero(((noisy == mask_id) & answer_region).reshape(-1)).reshape(-1).to(torch.int32).contiguous()
        n_rows = rows.numel()
        un_ptr = None
        if cfg_scale > 0.:                                                                      # :30-35
            batch = torch.cat([noisy, torch.where(prompt_region, mask_id, noisy)])
            km = torch.cat([keym, keym])
            lg = model.forward_rows(batch, km, torch.cat([rows, rows + R * Lmax]).contiguous())
            un_ptr = lg[n_rows:].data_ptr()
        else:
            lg = model.forward_rows(noisy, keym, rows)                                          # :38
        row_loss = torch.empty(n_rows, dtype=torch.float32, device=device)
        total = torch.zeros(1, dtype=torch.float64, device=device)
        ws = torch.empty(lib.llada_rows_workspace_bytes(n_rows), dtype=torch.uint8, device=device)
        pm_t = torch.from_numpy(p_mask).to(device)
        _lib.check(lib.llada_masked_ce(lg.data_ptr(), un_ptr, float(cfg_scale), rows.data_ptr(), n_rows, lg.shape[-1],
                                       noisy.data_ptr(), seqs.data_ptr(), R, Lmax, mask_id, pm_t.data_ptr(),
                                       1 if round_ce_bf16 else 0, row_loss.data_ptr(), total.data_ptr(), ws.data_ptr(),
                                       ws.numel(), stream))                                     # :72
        losses = row_loss.cpu().numpy().astype(np.float64)
        group_of_row = (rows.cpu().numpy() // Lmax) // batch_size
        for j, (i, it, *_rest) in enumerate(chunk):
            sums[i, it] = losses[group_of_row == j].sum()                                       # row order, float64
        g0 = g1
    return [- sum((sums[i] / batch_size).tolist()) / n_iter for i in range(len(pairs))]         # :73-77
