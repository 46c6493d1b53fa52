"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the evaluation-harness methods that sit
on the sampling / likelihood path: /root/reference/eval_llada.py:172-208 and :267-275.

Parity unpinned: eval_llada.py cannot be imported offline (it needs lm_eval), so these functions
are restated from the source text and are not checked against outputs of the reference.  Tie rule
of suffix_greedy_prediction: torch.sort(descending=True) is not stable; the lowest masked position
wins here.
"""
import numpy as np

from .sampler import cfg_combine


def softmax_f32(rows):
    """torch.softmax(logits.to(float32), dim=-1), eval_llada.py:187."""
    z = rows.astype(np.float32)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z, dtype=np.float32)
    return e / e.sum(axis=-1, keepdims=True, dtype=np.float32)


def get_logits(model, batch, prompt_index, cfg, mask_id):
    """eval_llada.py:136-149 (same arithmetic as get_log_likelihood.py:29-43)."""
    if cfg > 0.0:
        un = batch.copy()
        un[:, prompt_index] = mask_id
        lg = model(np.concatenate([batch, un], axis=0), None)
        n = batch.shape[0]
        return cfg_combine(lg[:n], lg[n:], cfg)
    return model(batch, None)


def suffix_greedy_prediction(model, prefix, target, cfg=0.0, mask_id=126336, trace=None):
    """eval_llada.py:173-194: fill the answer one token per step, always the masked position
    whose greedy prediction has the highest float32 softmax probability; True iff the filled
    answer equals ``target``."""
    l1, T = len(prefix), len(target)
    seq = np.full((1, l1 + T), mask_id, dtype=np.int64)                    # :177
    prompt_index = np.arange(l1 + T) < l1                                   # :178
    seq[0, :l1] = prefix                                                    # :180
    for _ in range(T):                                                      # :182
        mask_index = seq == mask_id                                         # :183
        logits = get_logits(model, seq, prompt_index, cfg, mask_id)[mask_index]   # :184
        x0 = logits.argmax(axis=-1)                                         # :185
        p = softmax_f32(logits)                                             # :187
        conf = p[np.arange(len(x0)), x0]                                    # :188
        keep = int(np.argmax(conf))                                         # :189 (first maximum)
        filled = np.full(len(x0), mask_id, dtype=np.int64)                  # :190
        filled[keep] = x0[keep]
        seq[mask_index] = filled                                            # :191
        if trace is not None:
            trace.append(seq[0].copy())
    return bool(np.all(np.asarray(target) == seq[0, l1:])), seq[0]


def encode_pair(tokenize, context, continuation):
    """eval_llada.py:196-208: trailing spaces of the context move to the continuation; the
    continuation ids are what the joint encoding adds after the context's own encoding."""
    n_spaces = len(context) - len(context.rstrip())
    if n_spaces > 0:
        continuation = context[-n_spaces:] + continuation
        context = context[:-n_spaces]
    whole = tokenize(context + continuation)
    ctx = tokenize(context)
    return ctx, whole[len(ctx):]


def trim_at_stops(text, stop_seqs):
    """eval_llada.py:268-270."""
    for s in stop_seqs:
        if s in text:
            text = text.split(s)[0]
    return text
