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
from .model import LLaDAModel
from .parallel import combine_loss_sums, shard_bounds


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
                       process_group=None):
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
    '''
    lib = _lib.load()
    device = torch.device(model.device)
    seq = torch.concatenate([prompt.to(device), answer.to(device)])[None, :]                    # :61
    seq = seq.repeat((batch_size, 1)).contiguous()                                              # :62
    l1, l = len(prompt), seq.shape[1]
    T = l - l1
    assert T >= 1
    n_iter = mc_num // batch_size                                                               # :66
    if n_iter == 0:
        raise ZeroDivisionError("division by zero")                                             # :77 with no iteration
    fast = isinstance(model, LLaDAModel)

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
        for it in range(n_iter):
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
    return combine_loss_sums(loss_sums, batch_size, world_size, process_group)
