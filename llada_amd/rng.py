"""Host-side scalar Philox4x32-10, bit-identical to csrc/philox.h.  Only used for the few host
decisions that must agree with the device streams (the per-iteration k of the likelihood
estimator, which the reference draws with torch.randint at get_log_likelihood.py:11)."""

_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF

STREAM_GUMBEL, STREAM_REMASK, STREAM_LL = 1, 2, 3


def philox4x32(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & _MASK, p1 >> 32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def key(seed, stream):
    seed &= 0xFFFFFFFFFFFFFFFF
    return seed & _MASK, ((seed >> 32) ^ stream) & _MASK


def ll_k(seed, iteration, target_len):
    """k in [1, target_len] for one Monte-Carlo iteration."""
    k0, k1 = key(seed, STREAM_LL)
    o0 = philox4x32(0xFFFFFFFF, 0, 0xFFFFFFFF, iteration & _MASK, k0, k1)[0]
    return o0 % target_len + 1
