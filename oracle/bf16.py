"""bfloat16 emulation on top of numpy float32 (test infrastructure).

A "bf16 value" is carried as a float32 whose low 16 mantissa bits are zero.
``round_bf16`` is round-to-nearest-even on the float32 bit pattern, which is what
torch's ``.to(torch.bfloat16)`` and the gfx950 ``v_cvt_pk_bf16_f32`` do.
"""
import numpy as np


def round_bf16(x):
    """float32 array -> float32 array holding the nearest bf16 values (RNE, NaN kept)."""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = ((u + np.uint64(0x7FFF) + lsb) >> np.uint64(16)) << np.uint64(16)
    out = (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
    nan = np.isnan(x)
    if np.any(nan):
        out = np.where(nan, np.float32(np.nan), out)
    return out.reshape(x.shape)


def to_bits(x):
    """bf16-valued float32 array -> uint16 bit patterns."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return (x.view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def from_bits(b):
    """uint16 bf16 bit patterns -> float32 array."""
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def ulp_distance(a, b):
    """Distance in bf16 ulps between two bf16-valued float32 arrays (same-sign finite values)."""
    ia = to_bits(a).astype(np.int64)
    ib = to_bits(b).astype(np.int64)
    # map sign-magnitude to a monotone integer line
    ia = np.where(ia & 0x8000, 0x8000 - ia, ia)
    ib = np.where(ib & 0x8000, 0x8000 - ib, ib)
    return np.abs(ia - ib)
