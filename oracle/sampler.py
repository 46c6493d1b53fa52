"""numpy restatement of the reference sampler (test infrastructure; see oracle/__init__.py).

Follows /root/reference/generate.py line by line; each function cites the lines it restates.
Tensors: ids are int64 ndarrays, logits are float32 ndarrays holding bf16 values (the
reference model emits bf16 logits), confidences are float32 ndarrays holding bf16 values
(or plain float32 for 'random' remasking, as in the reference).
"""
import numpy as np

from .bf16 import round_bf16

EOS_ID = 126081   # generate.py:95,101
EOT_ID = 126348   # generate.py:101
REMASKING = ("low_confidence", "high_confidence", "low_entropy", "high_entropy", "random")


def add_gumbel_noise(logits, temperature, u=None):
    """generate.py:8-19.  T == 0 returns the logits object itself; otherwise float64
    exp(logits) / (-log u)**T with u ~ U[0,1) float64 (passed in, the reference draws it
    with torch.rand_like)."""
    if temperature == 0:
        return logits
    l64 = logits.astype(np.float64)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        g = (-np.log(u.astype(np.float64))) ** np.float64(temperature)
        return np.exp(l64) / g


def get_num_transfer_tokens(mask_index, steps):
    """generate.py:22-40.  [B, n] bool -> [B, steps] int64 schedule."""
    mask_num = mask_index.sum(axis=1, keepdims=True).astype(np.int64)
    base = mask_num // steps
    rem = mask_num % steps
    out = np.zeros((mask_num.shape[0], steps), dtype=np.int64) + base
    for i in range(mask_num.shape[0]):
        out[i, : int(rem[i, 0])] += 1
    return out


def softmax_bf16(logits):
    """F.softmax on a bf16 tensor: float32 internally, result rounded to bf16."""
    with np.errstate(invalid="ignore", over="ignore"):
        m = np.max(logits, axis=-1, keepdims=True)
        e = np.exp((logits - m).astype(np.float32))
        s = e.sum(axis=-1, keepdims=True, dtype=np.float64).astype(np.float32)
        return round_bf16(e / s)


def log_softmax_bf16(logits, torch_cpu=False):
    """F.log_softmax on a bf16 tensor: float32 internally, result rounded to bf16.

    ``torch_cpu=True`` reproduces a quirk of torch's CPU kernel for bf16 inputs (the golden
    vectors were captured on CPU): it keeps the per-row sum and its logarithm in bf16
    scratch variables, i.e. lse = bf16(log(bf16(sum))).  GPU kernels keep both in float32,
    which is the default here and what the HIP path implements."""
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        m = np.max(logits, axis=-1, keepdims=True)
        z = (logits - m).astype(np.float32)
        s = np.exp(z).sum(axis=-1, keepdims=True, dtype=np.float64).astype(np.float32)
        if torch_cpu:
            lse = round_bf16(np.log(round_bf16(s)))
        else:
            lse = np.log(s).astype(np.float32)
        return round_bf16((z - lse).astype(np.float32))


def cfg_combine(cond, uncond, cfg_scale):
    """generate.py:89-90 (and get_log_likelihood.py:41-42): un + (cfg+1)*(c-un), every
    elementwise op rounding to bf16."""
    with np.errstate(invalid="ignore"):
        d = round_bf16(cond - uncond)
        d = round_bf16(d * np.float32(cfg_scale + 1))
        return round_bf16(uncond + d)


def remask_scores(logits, remasking, temperature=0.0, u=None, rand_scores=None,
                  logits_eos_inf=False, confidence_eos_eot_inf=False, torch_cpu=False):
    """generate.py:94-123 for one step.  logits [B, L, V] (bf16 values; modified in place
    exactly where the reference modifies them).  Returns (x0 [B, L] int64, x0_p [B, L])."""
    if logits_eos_inf:                                            # :94-95
        logits[:, :, EOS_ID] = -np.inf
    lwn = add_gumbel_noise(logits, temperature, u)               # :97
    x0 = np.argmax(lwn, axis=-1).astype(np.int64)                # :98 (first max on ties)
    if confidence_eos_eot_inf:                                    # :100-101 chained assignment
        lwn[:, :, EOS_ID] = -np.inf                               # same array as logits iff T == 0
        logits[:, :, EOT_ID] = -np.inf
    if remasking == "low_confidence":                             # :104-106
        p = softmax_bf16(logits)
        x0_p = np.take_along_axis(p, x0[..., None], axis=-1)[..., 0]
    elif remasking == "high_confidence":                          # :107-109
        p = round_bf16(np.float32(1) - softmax_bf16(logits))
        x0_p = np.take_along_axis(p, x0[..., None], axis=-1)[..., 0]
    elif remasking in ("low_entropy", "high_entropy"):            # :110-119
        log_p = log_softmax_bf16(logits, torch_cpu)
        p = softmax_bf16(logits)
        with np.errstate(invalid="ignore"):
            prod = round_bf16(log_p * p)
            s = round_bf16(prod.sum(axis=-1, dtype=np.float64).astype(np.float32))
        x0_p = -s if remasking == "low_entropy" else s
    elif remasking == "random":                                   # :120-121
        x0_p = np.asarray(rand_scores, dtype=np.float32).copy()
    else:
        raise NotImplementedError(remasking)                      # :122-123
    return x0, x0_p.astype(np.float32)


def topk_order(conf):
    """Indices of ``conf`` (1-D) sorted the way the build defines torch.topk's order:
    NaN first (torch ranks NaN highest), then larger values, ties by lower index."""
    idx = np.arange(conf.shape[0])
    isnan = np.isnan(conf)
    val = np.where(isnan, np.float32(0), conf)
    return np.lexsort((idx, -val.astype(np.float64), ~isnan))


def topk_transfer(x, x0, confidence, k):
    """generate.py:130-134.  In-place update of x; returns the bool transfer mask."""
    transfer = np.zeros_like(x, dtype=bool)
    for j in range(x.shape[0]):
        sel = topk_order(confidence[j])[: int(k[j])]
        transfer[j, sel] = True
    x[transfer] = x0[transfer]
    return transfer


def is_valid_topk(confidence_row, selected_mask, k):
    """True if ``selected_mask`` is one of the valid answers of topk(confidence_row, k):
    every value strictly above the k-th largest is selected, the rest tie with it."""
    k = int(k)
    if int(selected_mask.sum()) != k:
        return False
    if k == 0:
        return True
    order = topk_order(confidence_row)
    kth = confidence_row[order[k - 1]]
    key = np.where(np.isnan(confidence_row), np.inf, confidence_row)
    kth_key = np.inf if np.isnan(kth) else kth
    must = key > kth_key
    may = key >= kth_key
    return bool(np.all(selected_mask[must]) and np.all(may[selected_mask]))


def confidence_step(x, logits, mask_id, block_end, remasking, temperature=0.0, u=None, rand_scores=None,
                    logits_eos_inf=False, confidence_eos_eot_inf=False, torch_cpu=False):
    """generate.py:81,94-128 for one step: candidate tokens and confidences.
    x [B, L] int64, logits [B, L, V]; block_end = P + (num_block+1)*block_length."""
    mask_index = x == mask_id                                     # :81
    x0, x0_p = remask_scores(logits, remasking, temperature, u, rand_scores,
                             logits_eos_inf, confidence_eos_eot_inf, torch_cpu)
    x0_p[:, block_end:] = -np.inf                                 # :125
    x0 = np.where(mask_index, x0, x)                              # :127
    confidence = np.where(mask_index, x0_p, np.float32(-np.inf)).astype(np.float32)   # :128
    return x0, confidence


def generate(model, prompt, attention_mask=None, steps=128, gen_length=128, block_length=128,
             temperature=0.0, cfg_scale=0.0, remasking="low_confidence", mask_id=126336,
             logits_eos_inf=False, confidence_eos_eot_inf=False, *, noise_fn=None, rand_fn=None,
             trace=None, torch_cpu=False):
    """generate.py:43-136.  ``model(x, attention_mask)`` returns float32 logits holding bf16
    values.  ``noise_fn(shape)`` supplies the float64 uniforms of torch.rand_like (:17),
    ``rand_fn(shape)`` the float32 uniforms of torch.rand (:121).  ``trace`` (a list) receives
    one dict per step."""
    prompt = np.asarray(prompt, dtype=np.int64)
    B, P = prompt.shape
    x = np.full((B, P + gen_length), mask_id, dtype=np.int64)     # :60-61
    x[:, :P] = prompt
    if attention_mask is not None:                                # :63-64
        attention_mask = np.concatenate(
            [np.asarray(attention_mask), np.ones((B, gen_length), dtype=np.asarray(attention_mask).dtype)], axis=-1)
    prompt_index = x != mask_id                                   # :66
    assert gen_length % block_length == 0                         # :68
    num_blocks = gen_length // block_length
    assert steps % num_blocks == 0                                # :71
    steps = steps // num_blocks
    for nb in range(num_blocks):                                  # :74
        blk = x[:, P + nb * block_length: P + (nb + 1) * block_length] == mask_id
        num_transfer = get_num_transfer_tokens(blk, steps)        # :79
        for i in range(steps):                                    # :80
            if cfg_scale > 0.0:                                   # :82-90
                un_x = x.copy()
                un_x[prompt_index] = mask_id
                x_ = np.concatenate([x, un_x], axis=0)
                if attention_mask is None:
                    # generate.py:86-88 leaves attention_mask_ unbound in this case
                    raise UnboundLocalError("attention_mask_")
                am = np.concatenate([attention_mask, attention_mask], axis=0)
                lg = model(x_, am)
                logits = cfg_combine(lg[:B], lg[B:], cfg_scale)
            else:
                logits = np.array(model(x, attention_mask), dtype=np.float32, copy=True)   # :92
            u = None
            if temperature != 0:
                u = noise_fn(logits.shape)
            rs = rand_fn((B, x.shape[1])) if remasking == "random" else None
            x0, confidence = confidence_step(x, logits, mask_id, P + (nb + 1) * block_length, remasking,
                                             temperature, u, rs, logits_eos_inf, confidence_eos_eot_inf,
                                             torch_cpu)
            x_in = x.copy() if trace is not None else None
            transfer = topk_transfer(x, x0, confidence, num_transfer[:, i])   # :130-134
            if trace is not None:
                trace.append(dict(block=nb, step=i, x_in=x_in, x0=x0.copy(), confidence=confidence.copy(),
                                  k=num_transfer[:, i].copy(), transfer=transfer, x_out=x.copy()))
    return x                                                      # :136
