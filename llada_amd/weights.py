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


class LLaDAWeights:
    """Device-resident weights.  ``packed`` selects the layout above; ``packed=False`` keeps
    the checkpoint order (wqkv = q|k|v, wgu = gate|up) for the unfused reference path."""

    def __init__(self, cfg, device, packed=True):
        self.cfg, self.device, self.packed = cfg, torch.device(device), packed
        self.wte = self.ln_f = self.lm_head = None
        self.attn_norm, self.wqkv, self.wo, self.ff_norm, self.wgu, self.wdown = [], [], [], [], [], []

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def random(cls, cfg, device, seed=0, std=0.02, head_scale=8.0, packed=True):
        """i.i.d. N(0, std^2) bf16 weights at the configured shapes, norm weights 1, LM-head rows
        scaled by head_scale so logits are not degenerate.  Generated on the device."""
        self = cls(cfg, device, packed)
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)

        def w(n, k, s=std):
            return (torch.randn(n, k, generator=g, device=self.device, dtype=torch.float32) * s).to(torch.bfloat16)

        d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
        self.wte = w(V, d)
        ones = torch.ones(d, device=self.device, dtype=torch.bfloat16)
        for _ in range(cfg.n_layers):
            self.attn_norm.append(ones.clone())
            q, k, v = w(d, d), w(d, d), w(d, d)
            self.wqkv.append(pack_qkv(q, k, v, cfg.n_heads) if packed else torch.cat([q, k, v]).contiguous())
            del q, k, v
            self.wo.append(w(d, d))
            self.ff_norm.append(ones.clone())
            gate, up = w(F, d), w(F, d)
            self.wgu.append(pack_gate_up(gate, up) if packed else torch.cat([gate, up]).contiguous())
            del gate, up
            self.wdown.append(w(d, F))
        self.ln_f = ones.clone()
        self.lm_head = w(V, d, std * head_scale)
        return self

    @classmethod
    def from_numpy(cls, cfg, W, device, packed=True):
        """From a dict of float32 arrays holding bf16 values, keyed like the checkpoint
        (wte, blocks.{i}.{attn_norm,q_proj,...}, ln_f, ff_out)."""
        self = cls(cfg, device, packed)

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
        return self

    @classmethod
    def from_checkpoint(cls, path, device, cfg=None, packed=True):
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
        self = cls(cfg, device, packed)

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
        return self
