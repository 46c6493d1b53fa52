"""Weights of the mask predictor: configuration, the packed HBM layout the HIP GEMM epilogues
expect, random initialisation at the true LLaDA-8B shapes, and a local-checkpoint loader.

Layout in HBM (all bf16, row-major [out_features, in_features] like nn.Linear, so both GEMM
operands are K-contiguous):
  wqkv [3d, d]  = Q | K | V rows; inside every Q and K head the 128 rows are reordered
                  0-15,64-79,16-31,80-95,32-47,96-111,48-63,112-127 (rotary partners j, j+64 land in
                  neighbouring 16-column fragments of one MFMA lane; attention is invariant to a common
                  permutation of the Q/K head dimensions),
  wgu  [2F, d]  = gate/up rows interleaved in blocks of 16 (SwiGLU partners in one lane),
  wo [d, d], wdown [d, F], lm_head [V, d], wte [V, d] as in the checkpoint.
With ``tiled`` (the default when every projection has a multiple of 256 rows) the five projection matrices
are then stored tile-major: each 256-row x 64-column block contiguous, the blocks of one 256-row band
consecutive along K (``tile_major``).  One K step of a GEMM tile then reads one contiguous 32 KiB run of
HBM instead of 256 segments of 128 bytes, which is worth 2-10 % whenever the weights stream cold from
memory (every layer has its own, 436 MB per layer against 256 MB of Infinity Cache).
The reference loads the model with AutoModel.from_pretrained (generate.py:143,
get_log_likelihood.py:83); tensor names below follow that checkpoint.
"""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class LLaDAConfig:
    d_model: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    mlp_hidden: int = 12288
    vocab_size: int = 126464
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    mask_token_id: int = 126336

    @property
    def head_dim(self):
        return self.d_model // self.n_heads

    def linear_params(self):
        d, F, V = self.d_model, self.mlp_hidden, self.vocab_size
        return self.n_layers * (4 * d * d + 3 * d * F) + d * V


def head_permutation():
    """Row order inside one 128-wide Q or K head of the packed wqkv."""
    blocks = []
    for i in range(4):
        blocks.append(np.arange(16 * i, 16 * i + 16))
        blocks.append(np.arange(64 + 16 * i, 64 + 16 * i + 16))
    return np.concatenate(blocks)


def pack_qkv(q, k, v, n_heads):
    """q, k, v [d, d] torch tensors -> packed wqkv [3d, d]."""
    d = q.shape[0]
    perm = torch.from_numpy(head_permutation()).to(q.device)
    idx = (torch.arange(n_heads, device=q.device)[:, None] * 128 + perm[None, :]).reshape(-1)
    return torch.cat([q[idx], k[idx], v], dim=0).contiguous()


def pack_gate_up(gate, up):
    """gate, up [F, d] -> packed wgu [2F, d] interleaved in blocks of 16 rows."""
    F, d = gate.shape
    g = gate.reshape(F // 16, 16, d)
    u = up.reshape(F // 16, 16, d)
    return torch.stack([g, u], dim=1).reshape(2 * F, d).contiguous()


def tile_major(w):
    """[N, K] (N % 256 == 0, K % 64 == 0) -> the same shape, memory reordered so that element (n, k) sits at
    ((n // 256 * (K // 64) + k // 64) * 256 + n % 256) * 64 + k % 64."""
    N, K = w.shape
    assert N % 256 == 0 and K % 64 == 0
    return w.reshape(N // 256, 256, K // 64, 64).permute(0, 2, 1, 3).contiguous().reshape(N, K)


def from_tile_major(w):
    """Inverse of tile_major."""
    N, K = w.shape
    return w.reshape(N // 256, K // 64, 256, 64).permute(0, 2, 1, 3).contiguous().reshape(N, K)


def can_tile(cfg):
    d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
    return all(n % 256 == 0 for n in (d, 3 * d, 2 * F, V)) and d % 64 == 0 and F % 64 == 0


class LLaDAWeights:
    """Device-resident weights.  ``packed`` selects the layout above; ``packed=False`` keeps
    the checkpoint order (wqkv = q|k|v, wgu = gate|up) for the unfused reference path."""

    def __init__(self, cfg, device, packed=True, tiled=None):
        self.cfg, self.device, self.packed = cfg, torch.device(device), packed
        self.tiled = can_tile(cfg) if tiled is None else bool(tiled)
        assert not self.tiled or can_tile(cfg), "tile-major weights need every projection's rows to be a multiple of 256"
        self.wte = self.ln_f = self.lm_head = None
        self.attn_norm, self.wqkv, self.wo, self.ff_norm, self.wgu, self.wdown = [], [], [], [], [], []

    # -- constructors ------------------------------------------------------------------------
    def _finish(self):
        """Reorders the projection matrices in place (one at a time) once they are all loaded."""
        if self.tiled:
            for lst in (self.wqkv, self.wo, self.wgu, self.wdown):
                for i in range(len(lst)):
                    lst[i] = tile_major(lst[i])
            self.lm_head = tile_major(self.lm_head)
        return self

    @classmethod
    def random(cls, cfg, device, seed=0, std=0.02, head_scale=8.0, packed=True, tiled=None, resid_std=None, embed_std=None):
        """i.i.d. N(0, std^2) bf16 weights at the configured shapes, norm weights 1, LM-head rows
        scaled by head_scale so logits are not degenerate.  Generated on the device.
        ``resid_std``: standard deviation of the two matrices that write into the residual stream (attn_out,
        ff_out); None = std.  ``std / sqrt(2 * n_layers)`` is the depth-scaled initialisation of trained
        Llama / OLMo-style stacks: every layer then moves the stream by a small fraction of its size and a
        rounding made in one layer is not amplified by the ones above it.  The generator stream is the same
        for every value (the draw is scaled afterwards).  ``embed_std``: standard deviation of the embedding
        table (None = std).  A pre-norm stack is invariant to the common scale of everything written into the
        stream, so what conditions it is the RATIO embedding : layer update; embed_std >> resid_std * sqrt(width)
        gives a stream every layer changes only slightly."""
        self = cls(cfg, device, packed, tiled)
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)

        def w(n, k, s=std):
            return (torch.randn(n, k, generator=g, device=self.device, dtype=torch.float32) * s).to(torch.bfloat16)

        d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
        rs = std if resid_std is None else resid_std
        self.wte = w(V, d, std if embed_std is None else embed_std)
        ones = torch.ones(d, device=self.device, dtype=torch.bfloat16)
        for _ in range(cfg.n_layers):
            self.attn_norm.append(ones.clone())
            q, k, v = w(d, d), w(d, d), w(d, d)
            self.wqkv.append(pack_qkv(q, k, v, cfg.n_heads) if packed else torch.cat([q, k, v]).contiguous())
            del q, k, v
            self.wo.append(w(d, d, rs))
            self.ff_norm.append(ones.clone())
            gate, up = w(F, d), w(F, d)
            self.wgu.append(pack_gate_up(gate, up) if packed else torch.cat([gate, up]).contiguous())
            del gate, up
            self.wdown.append(w(d, F, rs))
        self.ln_f = ones.clone()
        self.lm_head = w(V, d, std * head_scale)
        return self._finish()

    @classmethod
    def from_numpy(cls, cfg, W, device, packed=True, tiled=None):
        """From a dict of float32 arrays holding bf16 values, keyed like the checkpoint
        (wte, blocks.{i}.{attn_norm,q_proj,...}, ln_f, ff_out)."""
        self = cls(cfg, device, packed, tiled)

        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.device).to(torch.bfloat16)

        self.wte = t(W["wte"])
        for i in range(cfg.n_layers):
            p = f"blocks.{i}."
            self.attn_norm.append(t(W[p + "attn_norm"]))
            q, k, v = t(W[p + "q_proj"]), t(W[p + "k_proj"]), t(W[p + "v_proj"])
            self.wqkv.append(pack_qkv(q, k, v, cfg.n_heads) if packed else torch.cat([q, k, v]).contiguous())
            self.wo.append(t(W[p + "attn_out"]))
            self.ff_norm.append(t(W[p + "ff_norm"]))
            gate, up = t(W[p + "ff_proj"]), t(W[p + "up_proj"])
            self.wgu.append(pack_gate_up(gate, up) if packed else torch.cat([gate, up]).contiguous())
            self.wdown.append(t(W[p + "ff_out"]))
        self.ln_f = t(W["ln_f"])
        self.lm_head = t(W["ff_out"])
        return self._finish()

    @classmethod
    def from_checkpoint(cls, path, device, cfg=None, packed=True, tiled=None):
        """Load a local LLaDA checkpoint directory (*.safetensors, names
        model.transformer.{wte,ln_f,ff_out}.weight and model.transformer.blocks.{i}.*.weight)."""
        import glob
        import json
        import os

        from safetensors import safe_open
        if cfg is None:
            with open(os.path.join(path, "config.json")) as f:
                c = json.load(f)
            cfg = LLaDAConfig(d_model=c["d_model"], n_layers=c["n_layers"], n_heads=c["n_heads"],
                              mlp_hidden=c.get("mlp_hidden_size", 12288),
                              vocab_size=c.get("embedding_size", c.get("vocab_size", 126464)),
                              rope_theta=c.get("rope_theta", 500000.0), rms_eps=c.get("rms_norm_eps", 1e-5),
                              mask_token_id=c.get("mask_token_id", 126336))
        tensors = {}
        for fn in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
            with safe_open(fn, framework="pt", device="cpu") as f:
                for name in f.keys():
                    tensors[name.replace("model.transformer.", "").replace(".weight", "")] = f.get_tensor(name)
        self = cls(cfg, device, packed, tiled)

        def t(name):
            return tensors[name].to(self.device).to(torch.bfloat16)

        self.wte = t("wte")
        for i in range(cfg.n_layers):
            p = f"blocks.{i}."
            self.attn_norm.append(t(p + "attn_norm"))
            q, k, v = t(p + "q_proj"), t(p + "k_proj"), t(p + "v_proj")
            self.wqkv.append(pack_qkv(q, k, v, cfg.n_heads) if packed else torch.cat([q, k, v]).contiguous())
            self.wo.append(t(p + "attn_out"))
            self.ff_norm.append(t(p + "ff_norm"))
            gate, up = t(p + "ff_proj"), t(p + "up_proj")
            self.wgu.append(pack_gate_up(gate, up) if packed else torch.cat([gate, up]).contiguous())
            self.wdown.append(t(p + "ff_out"))
        self.ln_f = t("ln_f")
        self.lm_head = t("ff_out")
        return self._finish()
