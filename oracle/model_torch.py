"""torch-CPU restatement of the LLaDA-8B non-causal forward: the arithmetic of oracle/model.py (same
architecture, same places where a bf16 model run rounds) on torch's multi-threaded CPU kernels, i.e. what
the reference's callers execute when they run `model(x).logits` (generate.py:88,92) on a CPU in
`torch_dtype=torch.bfloat16`.  TEST INFRASTRUCTURE, like the rest of oracle/: used by bench.py's
`cpu_baseline` leg (the reference's CPU path timed on the host) and checked against oracle/model.py in
tests/; never imported by the product package.  PARITY UNPINNED against the reference for the same reason
as oracle/model.py (the model code is not in the reference tree).

Weights: dict of torch bf16 tensors with the checkpoint's names (see oracle/model.py).  ``layer_of(i)``
maps a layer index to the index whose weights it uses, so a timing run can cycle a few distinct weight
sets through all 32 layers without holding 15 GB (same FLOPs and bytes per layer; the sets are far larger
than any CPU cache).
"""
import torch


def linear(x, w, mode):
    """x [.., K] bf16 @ w [N, K] -> bf16.  mode "bf16": torch's native bf16 CPU GEMM (float32 accumulation
    inside oneDNN); mode "f32": float32 GEMM on upcast operands, result rounded to bf16 (a weight already
    held in float32 -- bf16 values, upcast once by the caller -- is used as it is)."""
    if mode == "bf16":
        return torch.nn.functional.linear(x, w)
    return torch.nn.functional.linear(x.float(), w if w.dtype == torch.float32 else w.float()).to(torch.bfloat16)


def rmsnorm(x, w, eps):
    x32 = x.float()
    xn = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16)
    return (xn.float() * w.float()).to(torch.bfloat16)


def rope_tables(L, head_dim, theta):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    ang = torch.arange(L, dtype=torch.float32)[:, None] * inv.float()[None, :]
    return torch.cos(ang), torch.sin(ang)


def apply_rope(t, cos, sin):
    """t [B, H, L, D] bf16; half-split rotation in float32, result bf16."""
    half = t.shape[-1] // 2
    t32 = t.float()
    t1, t2 = t32[..., :half], t32[..., half:]
    c, s = cos[None, None], sin[None, None]
    return torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], dim=-1).to(torch.bfloat16)


@torch.no_grad()
def forward(W, cfg, ids, attention_mask=None, mode="f32", layer_of=None, logit_positions=None):
    """ids int64 [B, L] -> logits bf16 [B, L, V] (or [B, len(logit_positions), V])."""
    B, L = ids.shape
    H, D = cfg.n_heads, cfg.head_dim
    x = W["wte"][ids % W["wte"].shape[0]].to(torch.bfloat16)   # timing runs may carry a folded embedding table
    cos, sin = rope_tables(L, D, cfg.rope_theta)
    bias = None
    if attention_mask is not None:
        bias = torch.zeros(B, 1, 1, L, dtype=torch.bfloat16)
        bias.masked_fill_(attention_mask[:, None, None, :] == 0, float("-inf"))
    for i in range(cfg.n_layers):
        p = f"blocks.{i if layer_of is None else layer_of(i)}."
        h = rmsnorm(x, W[p + "attn_norm"], cfg.rms_eps)
        q = linear(h, W[p + "q_proj"], mode).view(B, L, H, D).transpose(1, 2)
        k = linear(h, W[p + "k_proj"], mode).view(B, L, H, D).transpose(1, 2)
        v = linear(h, W[p + "v_proj"], mode).view(B, L, H, D).transpose(1, 2)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        a = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias, is_causal=False)
        a = a.transpose(1, 2).reshape(B, L, H * D)
        x = x + linear(a, W[p + "attn_out"], mode)
        h = rmsnorm(x, W[p + "ff_norm"], cfg.rms_eps)
        g = linear(h, W[p + "ff_proj"], mode)
        u = linear(h, W[p + "up_proj"], mode)
        x = x + linear(torch.nn.functional.silu(g) * u, W[p + "ff_out"], mode)
    hf = rmsnorm(x, W["ln_f"], cfg.rms_eps)
    if logit_positions is not None:
        hf = hf[:, logit_positions]
    return linear(hf, W["ff_out"], mode)


def from_numpy_weights(Wnp):
    """oracle.model.init_weights() dict (float32 arrays holding bf16 values) -> torch bf16 tensors."""
    return {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in Wnp.items()}


def pick_mode(d=4096, rows=192):
    """The faster of the two GEMM forms on this host (native bf16 needs AMX / AVX512-BF16 to be fast)."""
    import time
    x = torch.randn(rows, d).to(torch.bfloat16)
    w16 = torch.randn(d, d).to(torch.bfloat16)
    w32 = w16.float()
    best = None
    for mode, w in (("bf16", w16), ("f32", w32)):
        linear(x, w, mode)
        t0 = time.perf_counter()
        for _ in range(3):
            linear(x, w, mode)
        t = time.perf_counter() - t0
        if best is None or t < best[1]:
            best = (mode, t)
    return best[0]
