"""numpy restatement of the reference likelihood estimator (test infrastructure; see
oracle/__init__.py).  Follows /root/reference/get_log_likelihood.py; the same arithmetic is
duplicated in /root/reference/eval_llada.py:113-170.
"""
import numpy as np

from .bf16 import round_bf16
from .sampler import cfg_combine, log_softmax_bf16


def stratified_counts(k, b, target_len):
    """get_log_likelihood.py:13-14.  float32 linspace(k, k+(b-1)*(T/b), b) rounded half-to-even,
    then ((x-1) mod T) + 1.  torch.linspace on CPU computes start + i*step for the first half
    and end - (steps-1-i)*step for the second half, in float32."""
    start = np.float32(float(k))
    end = np.float32(k + (b - 1) * (target_len / b))
    if b == 1:
        vals = np.array([start], dtype=np.float32)
    else:
        step = np.float32((end - start) / np.float32(b - 1))
        i = np.arange(b)
        half = b // 2
        lo = (start + step * i.astype(np.float32)).astype(np.float32)
        hi = (end - step * (b - 1 - i).astype(np.float32)).astype(np.float32)
        vals = np.where(i < half, lo, hi).astype(np.float32)
    x = np.rint(vals).astype(np.int64)                 # torch.round = half to even
    return ((x - 1) % target_len) + 1


def forward_process(batch, prompt_len, mask_id, k, perms):
    """get_log_likelihood.py:7-26.  ``k`` is the torch.randint draw (:11), ``perms`` [b, T] the
    per-row torch.randperm draws (:20).  Returns (noisy_batch int64 [b,l], p_mask float32 [b,l])."""
    b, l = batch.shape
    target_len = l - prompt_len
    x = stratified_counts(k, b, target_len)
    assert x.min() >= 1 and x.max() <= target_len                          # :15
    is_mask = np.arange(target_len)[None, :] < x[:, None]                   # :17-18
    for i in range(b):                                                      # :19-20
        is_mask[i] = is_mask[i][perms[i]]
    is_mask = np.concatenate([np.zeros((b, prompt_len), dtype=bool), is_mask], axis=1)   # :22
    noisy = np.where(is_mask, mask_id, batch)                               # :23
    p_mask = np.repeat((x.astype(np.float32) / np.float32(target_len))[:, None], l, axis=1)   # :26
    return noisy, p_mask.astype(np.float32)


def get_logits(model, batch, prompt_index, cfg_scale, mask_id):
    """get_log_likelihood.py:29-43."""
    if cfg_scale > 0.0:
        assert len(prompt_index) == batch.shape[1]                          # :31
        un = batch.copy()
        un[:, prompt_index] = mask_id
        both = np.concatenate([batch, un], axis=0)
        lg = model(both, None)
        n = batch.shape[0]
        return cfg_combine(lg[:n], lg[n:], cfg_scale)
    return model(batch, None)


def masked_ce(logits_rows, targets, round_to_bf16=True, torch_cpu=False):
    """F.cross_entropy(reduction='none') on bf16 logits rows [N, V]: float32 log-softmax, result
    rounded to bf16 (torch returns the loss in the logits' dtype).  ``torch_cpu``: see
    oracle.sampler.log_softmax_bf16."""
    if round_to_bf16:
        lp = log_softmax_bf16(logits_rows, torch_cpu)
        return (-lp[np.arange(lp.shape[0]), targets]).astype(np.float32)
    m = logits_rows.max(axis=-1, keepdims=True)
    z = (logits_rows - m).astype(np.float32)
    lse = np.log(np.exp(z).sum(axis=-1, dtype=np.float64))
    return (lse - z[np.arange(z.shape[0]), targets].astype(np.float64)).astype(np.float32)


def get_log_likelihood(model, prompt, answer, mc_num=128, batch_size=16, cfg_scale=0.0, mask_id=126336,
                       *, k_fn=None, perm_fn=None, round_to_bf16=True, trace=None,
                       torch_cpu=False):
    """get_log_likelihood.py:46-77.  ``k_fn(target_len)`` supplies torch.randint(1, T+1),
    ``perm_fn(target_len)`` each torch.randperm(T) in row order."""
    prompt = np.asarray(prompt, dtype=np.int64)
    answer = np.asarray(answer, dtype=np.int64)
    seq = np.repeat(np.concatenate([prompt, answer])[None, :], batch_size, axis=0)   # :61-62
    l = seq.shape[1]
    prompt_index = np.arange(l) < len(prompt)                                         # :63
    target_len = l - len(prompt)
    losses = []
    for it in range(mc_num // batch_size):                                            # :66
        k = k_fn(target_len)
        perms = np.stack([perm_fn(target_len) for _ in range(batch_size)])
        perturbed, p_mask = forward_process(seq, len(prompt), mask_id, k, perms)      # :67
        mask_index = perturbed == mask_id                                             # :68
        logits = get_logits(model, perturbed, prompt_index, cfg_scale, mask_id)       # :70
        ce = masked_ce(logits[mask_index], seq[mask_index], round_to_bf16, torch_cpu)          # :72
        loss = (ce.astype(np.float32) / p_mask[mask_index]).astype(np.float32)
        val = float(np.float32(loss.sum(dtype=np.float64)) / np.float32(batch_size))  # :73
        losses.append(val)                                                            # :75
        if trace is not None:
            trace.append(dict(k=k, perms=perms, perturbed=perturbed, p_mask=p_mask, loss=val))
    return -sum(losses) / len(losses)                                                 # :77
