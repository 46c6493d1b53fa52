"""numpy restatement of the LLaDA-8B non-causal forward (test infrastructure; see
oracle/__init__.py -- PARITY UNPINNED against the reference: the reference loads the model
code from the HF hub (generate.py:143, get_log_likelihood.py:83) and does not vendor it).

Architecture restated from /root/reference/GUIDELINES.md:6-16 (a Llama-style transformer with
the causal mask removed) and the published LLaDA-8B configuration: pre-norm blocks,
RMSNorm (float32 inside, weight applied after the cast back), rotary embeddings in float32
(half-split rotate, positions 0..L-1), plain multi-head attention over the full sequence with
an optional key-padding mask, SwiGLU MLP, no biases, untied LM head.

``emulate_bf16=True`` rounds to bf16 wherever a bf16 model run would (every linear output,
norm output, rotary output, attention output, residual sum), so the result is comparable with
the HIP path to bf16 tolerance.  Weight names follow the checkpoint:
``wte, blocks.{i}.{attn_norm,q_proj,k_proj,v_proj,attn_out,ff_norm,ff_proj,up_proj,ff_out}, ln_f, ff_out``.
"""
from dataclasses import dataclass

import numpy as np

from .bf16 import round_bf16


@dataclass
class LLaDAConfig:
    d_model: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    mlp_hidden: int = 12288
    vocab_size: int = 126464
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5

    @property
    def head_dim(self):
        return self.d_model // self.n_heads


def init_weights(cfg, seed=0, std=0.02, head_scale=1.0):
    """Random bf16-valued float32 weights from one PCG64 stream, in a fixed documented order
    (wte; per block q,k,v,attn_out,ff_proj,up_proj,ff_out; LM head).  Norm weights are 1."""
    g = np.random.Generator(np.random.PCG64(seed))

    def w(n, k, s=std):
        return round_bf16(g.standard_normal(size=(n, k), dtype=np.float32) * np.float32(s))

    W = {"wte": w(cfg.vocab_size, cfg.d_model)}
    for i in range(cfg.n_layers):
        p = f"blocks.{i}."
        W[p + "attn_norm"] = np.ones(cfg.d_model, dtype=np.float32)
        W[p + "q_proj"] = w(cfg.d_model, cfg.d_model)
        W[p + "k_proj"] = w(cfg.d_model, cfg.d_model)
        W[p + "v_proj"] = w(cfg.d_model, cfg.d_model)
        W[p + "attn_out"] = w(cfg.d_model, cfg.d_model)
        W[p + "ff_norm"] = np.ones(cfg.d_model, dtype=np.float32)
        W[p + "ff_proj"] = w(cfg.mlp_hidden, cfg.d_model)
        W[p + "up_proj"] = w(cfg.mlp_hidden, cfg.d_model)
        W[p + "ff_out"] = w(cfg.d_model, cfg.mlp_hidden)
    W["ln_f"] = np.ones(cfg.d_model, dtype=np.float32)
    W["ff_out"] = w(cfg.vocab_size, cfg.d_model, std * head_scale)
    return W


def _r(x, emulate):
    return round_bf16(x) if emulate else x.astype(np.float32)


def rmsnorm(x, w, eps, emulate):
    x32 = x.astype(np.float32)
    var = np.mean(x32.astype(np.float64) ** 2, axis=-1, keepdims=True).astype(np.float32)
    xn = _r(x32 * (np.float32(1.0) / np.sqrt(var + np.float32(eps))), emulate)
    return _r(xn * w.astype(np.float32), emulate)


def rope_tables(L, head_dim, theta):
    inv = 1.0 / (theta ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))
    # angles in float32 (position * inv_freq), as a float32 torch run builds them
    ang32 = (np.arange(L, dtype=np.float32)[:, None] * inv.astype(np.float32)[None, :]).astype(np.float32)
    return np.cos(ang32).astype(np.float32), np.sin(ang32).astype(np.float32)


def apply_rope(t, cos, sin, emulate):
    """t [B, H, L, D]; half-split rotation: (t1, t2) -> (t1*cos - t2*sin, t2*cos + t1*sin).  cos / sin [L, D/2]
    (the same positions for every sequence) or [B, L, D/2] (one table row per token: explicit positions)."""
    half = t.shape[-1] // 2
    t1, t2 = t[..., :half], t[..., half:]
    c = cos[None, None, :, :] if cos.ndim == 2 else cos[:, None, :, :]
    s = sin[None, None, :, :] if sin.ndim == 2 else sin[:, None, :, :]
    out = np.concatenate([t1 * c - t2 * s, t2 * c + t1 * s], axis=-1).astype(np.float32)
    return _r(out, emulate)


def attention(q, k, v, key_mask, emulate):
    """softmax(q k^T / sqrt(D)) v over the whole sequence; key_mask [B, L] (1 = attend)."""
    D = q.shape[-1]
    s = np.einsum("bhqd,bhkd->bhqk", q, k, optimize=True).astype(np.float32) * np.float32(1.0 / np.sqrt(D))
    if key_mask is not None:
        s = np.where(key_mask[:, None, None, :] != 0, s, np.float32(-np.inf))
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(axis=-1, keepdims=True)
    o = np.einsum("bhqk,bhkd->bhqd", p.astype(np.float32), v, optimize=True).astype(np.float32)
    return _r(o, emulate)


def block(x, Wl, cfg, cos, sin, attention_mask=None, emulate_bf16=True, kv_out=None):
    """One pre-norm transformer block on x [B, L, d] -> [B, L, d].  ``Wl`` holds the block's weights under the
    bare names attn_norm, q_proj, k_proj, v_proj, attn_out, ff_norm, ff_proj, up_proj, ff_out."""
    em = emulate_bf16
    B, L, _ = x.shape
    H, D = cfg.n_heads, cfg.head_dim
    h = rmsnorm(x, Wl["attn_norm"], cfg.rms_eps, em)
    q = _r(h @ Wl["q_proj"].T, em).reshape(B, L, H, D).transpose(0, 2, 1, 3)
    k = _r(h @ Wl["k_proj"].T, em).reshape(B, L, H, D).transpose(0, 2, 1, 3)
    v = _r(h @ Wl["v_proj"].T, em).reshape(B, L, H, D).transpose(0, 2, 1, 3)
    q = apply_rope(q, cos, sin, em)
    k = apply_rope(k, cos, sin, em)
    if kv_out is not None:
        kv_out.append((k.copy(), v.copy()))
    a = attention(q, k, v, attention_mask, em).transpose(0, 2, 1, 3).reshape(B, L, H * D)
    x = _r(x + _r(a @ Wl["attn_out"].T, em), em)
    h = rmsnorm(x, Wl["ff_norm"], cfg.rms_eps, em)
    g = _r(h @ Wl["ff_proj"].T, em)
    u = _r(h @ Wl["up_proj"].T, em)
    act = _r(g / (np.float32(1.0) + np.exp(-g)), em)     # silu
    return _r(x + _r(_r(act * u, em) @ Wl["ff_out"].T, em), em)


LAYER_KEYS = ("attn_norm", "q_proj", "k_proj", "v_proj", "attn_out", "ff_norm", "ff_proj", "up_proj", "ff_out")


def positions_from_mask(attention_mask):
    """Rotary positions counted from every row's first attended token: cumsum(mask) - 1, clamped at 0."""
    m = (np.asarray(attention_mask) != 0).astype(np.int64)
    return np.maximum(np.cumsum(m, axis=-1) - 1, 0)


def forward(W, cfg, ids, attention_mask=None, emulate_bf16=True, n_layers=None, return_hidden=False, kv_out=None,
            position_ids=None):
    """ids int64 [B, L] -> logits float32 [B, L, V] (bf16 values when emulate_bf16).  ``kv_out`` (a list)
    receives every layer's rotated keys and values (k, v), each [B, H, L, D], for forward_block.
    ``position_ids`` int [B, L]: rotary position of every token (default: its column, as the reference's model)."""
    em = emulate_bf16
    ids = np.asarray(ids)
    B, L = ids.shape
    x = W["wte"][ids].astype(np.float32)                     # [B, L, d]
    cos, sin = rope_tables(L, cfg.head_dim, cfg.rope_theta)
    if position_ids is not None:
        cos, sin = cos[np.asarray(position_ids)], sin[np.asarray(position_ids)]      # [B, L, D/2]
    nl = cfg.n_layers if n_layers is None else n_layers
    for i in range(nl):
        p = f"blocks.{i}."
        x = block(x, {k: W[p + k] for k in LAYER_KEYS}, cfg, cos, sin, attention_mask, em, kv_out)
    hf = rmsnorm(x, W["ln_f"], cfg.rms_eps, em)
    if return_hidden:
        return hf
    return _r(hf @ W["ff_out"].T, em)


def forward_block(W, cfg, block_ids, block_start, kv, attention_mask=None, emulate_bf16=True):
    """The block-cached step (no counterpart in the reference, whose sampler recomputes every position,
    generate.py:80-92): only the positions block_start .. block_start + bl - 1 run through the layers.
    Their keys and values replace the cached ones in ``kv`` (the list forward(kv_out=) filled, updated in
    place); their queries attend over the whole cached sequence.  block_ids int64 [B, bl] -> logits
    float32 [B, bl, V]; attention_mask is the full [B, L] key mask."""
    em = emulate_bf16
    block_ids = np.asarray(block_ids)
    B, bl = block_ids.shape
    H, D = cfg.n_heads, cfg.head_dim
    L = kv[0][0].shape[2]
    cos, sin = rope_tables(L, D, cfg.rope_theta)
    cos, sin = cos[block_start:block_start + bl], sin[block_start:block_start + bl]
    x = W["wte"][block_ids].astype(np.float32)
    for i in range(cfg.n_layers):
        p = f"blocks.{i}."
        h = rmsnorm(x, W[p + "attn_norm"], cfg.rms_eps, em)
        q = _r(h @ W[p + "q_proj"].T, em).reshape(B, bl, H, D).transpose(0, 2, 1, 3)
        k = _r(h @ W[p + "k_proj"].T, em).reshape(B, bl, H, D).transpose(0, 2, 1, 3)
        v = _r(h @ W[p + "v_proj"].T, em).reshape(B, bl, H, D).transpose(0, 2, 1, 3)
        q = apply_rope(q, cos, sin, em)
        k = apply_rope(k, cos, sin, em)
        kc, vc = kv[i]
        kc[:, :, block_start:block_start + bl] = k
        vc[:, :, block_start:block_start + bl] = v
        a = attention(q, kc, vc, attention_mask, em).transpose(0, 2, 1, 3).reshape(B, bl, H * D)
        x = _r(x + _r(a @ W[p + "attn_out"].T, em), em)
        h = rmsnorm(x, W[p + "ff_norm"], cfg.rms_eps, em)
        g = _r(h @ W[p + "ff_proj"].T, em)
        u = _r(h @ W[p + "up_proj"].T, em)
        act = _r(g / (np.float32(1.0) + np.exp(-g)), em)     # silu
        x = _r(x + _r(_r(act * u, em) @ W[p + "ff_out"].T, em), em)
    hf = rmsnorm(x, W["ln_f"], cfg.rms_eps, em)
    return _r(hf @ W["ff_out"].T, em)


class OracleModel:
    """Model-protocol wrapper (generate.py:60,88,92): callable (ids, attention_mask) -> logits."""

    def __init__(self, W, cfg, emulate_bf16=True):
        self.W, self.cfg, self.emulate = W, cfg, emulate_bf16

    def __call__(self, ids, attention_mask=None):
        return forward(self.W, self.cfg, ids, attention_mask, self.emulate)
