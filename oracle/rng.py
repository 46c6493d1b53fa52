"""Random streams shared by the oracle, the golden-vector tool and the HIP path (test infrastructure).

Two families:

1. ``fake_logits`` -- deterministic stand-in model outputs for the sampler tests:
   ``np.random.Generator(np.random.PCG64(seed + call_index))`` standard normals (float32),
   scaled and rounded to bf16.  Regenerated on both sides, never stored.

2. Philox4x32-10 counter RNG -- the product's own device RNG (Gumbel noise, 'random'
   remasking scores, likelihood masks).  The numpy version here is the checker for the HIP
   implementation in ``llada_amd/csrc/philox.h``.  Layout of one draw:

       counter = (c0, c1, c2, c3), key = (seed_lo, seed_hi ^ stream)
       Gumbel uniforms   : stream 1, c0 = vocab_index >> 1, c1 = position, c2 = global row,
                           c3 = global step; outputs (o0,o1) -> u[even v], (o2,o3) -> u[odd v]
       remask 'random'   : stream 2, c0 = position, c1 = 0, c2 = global row, c3 = global step;
                           score = (o0 >> 8) * 2^-24  (float32 in [0,1))
       likelihood masks  : stream 3, c0 = target position, c1 = 0, c2 = global row,
                           c3 = MC iteration; sort key = o0 (ties by position);
                           the iteration's k uses c0 = 0xFFFFFFFF, c2 = 0xFFFFFFFF.
   A float64 uniform is ((a >> 5) * 2^26 + (b >> 6)) * 2^-53 from two 32-bit words (a, b).
"""
import numpy as np

from .bf16 import round_bf16

STREAM_GUMBEL = 1
STREAM_REMASK = 2
STREAM_LL = 3

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def fake_logits(seed, call_index, shape, scale=3.0):
    g = np.random.Generator(np.random.PCG64(int(seed) + int(call_index)))
    x = g.standard_normal(size=shape, dtype=np.float32) * np.float32(scale)
    return round_bf16(x)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counters are broadcastable uint arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3)))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _key(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, ((seed >> 32) ^ stream) & 0xFFFFFFFF


def u64_from_words(a, b):
    a = a.astype(np.uint64)
    b = b.astype(np.uint64)
    return ((a >> np.uint64(5)) * np.float64(67108864.0) + (b >> np.uint64(6)).astype(np.float64)) * np.float64(2.0 ** -53)


def gumbel_uniforms(seed, step, rows, positions, vocab):
    """float64 uniforms [len(rows), len(positions), vocab] of the product's Gumbel stream."""
    k0, k1 = _key(seed, STREAM_GUMBEL)
    rows = np.asarray(rows, dtype=np.uint64)[:, None, None]
    pos = np.asarray(positions, dtype=np.uint64)[None, :, None]
    half = np.arange((vocab + 1) // 2, dtype=np.uint64)[None, None, :]
    o0, o1, o2, o3 = philox4x32(half, pos, rows, np.uint64(step), k0, k1)
    ue = u64_from_words(o0, o1)
    uo = u64_from_words(o2, o3)
    u = np.stack([ue, uo], axis=-1).reshape(ue.shape[0], ue.shape[1], -1)
    return u[:, :, :vocab]


def remask_uniforms(seed, step, rows, positions):
    """float32 scores [len(rows), len(positions)] of the product's 'random' remasking stream."""
    k0, k1 = _key(seed, STREAM_REMASK)
    rows = np.asarray(rows, dtype=np.uint64)[:, None]
    pos = np.asarray(positions, dtype=np.uint64)[None, :]
    o0, _, _, _ = philox4x32(pos, np.uint64(0), rows, np.uint64(step), k0, k1)
    return ((o0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def ll_sort_keys(seed, iteration, rows, target_len):
    """uint32 sort keys [len(rows), target_len] of the product's likelihood-mask stream."""
    k0, k1 = _key(seed, STREAM_LL)
    rows = np.asarray(rows, dtype=np.uint64)[:, None]
    pos = np.arange(target_len, dtype=np.uint64)[None, :]
    o0, _, _, _ = philox4x32(pos, np.uint64(0), rows, np.uint64(iteration), k0, k1)
    return o0


def ll_k(seed, iteration, target_len):
    """The iteration's k in [1, target_len] of the product's likelihood-mask stream."""
    k0, k1 = _key(seed, STREAM_LL)
    o0, _, _, _ = philox4x32(np.uint64(0xFFFFFFFF), np.uint64(0), np.uint64(0xFFFFFFFF), np.uint64(iteration), k0, k1)
    return int(o0) % int(target_len) + 1
