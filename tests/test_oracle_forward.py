"""Pins the oracle's forward restatement against the stand-in vectors (transformers' Llama run
bidirectionally; the reference's own model code is not vendored -- see oracle/model.py)."""
import numpy as np

from oracle.model import LLaDAConfig, forward, init_weights
from tests.golden_util import load


def _setup():
    z, _ = load("forward_standin.npz")
    d, nl, nh, ff, V = (int(v) for v in z["cfg"])
    cfg = LLaDAConfig(d_model=d, n_layers=nl, n_heads=nh, mlp_hidden=ff, vocab_size=V)
    W = init_weights(cfg, seed=int(z["weight_seed"][0]), std=float(z["weight_std"][0]),
                     head_scale=float(z["head_scale"][0]))
    return z, cfg, W


def test_forward_float32_matches_standin():
    z, cfg, W = _setup()
    out = forward(W, cfg, z["ids"], None, emulate_bf16=False)
    ref = z["logits_nomask"]
    assert np.abs(out - ref).max() <= 2e-4 * np.abs(ref).max()
    out = forward(W, cfg, z["ids"], z["key_mask"], emulate_bf16=False)
    ref = z["logits_padmask"]
    assert np.abs(out - ref).max() <= 2e-4 * np.abs(ref).max()


def test_forward_bf16_emulation_within_contract():
    """bf16 rounding points keep the logits within the 2e-2 max-rel contract of the float32 run."""
    z, cfg, W = _setup()
    out = forward(W, cfg, z["ids"], z["key_mask"], emulate_bf16=True)
    ref = z["logits_padmask"]
    assert np.abs(out - ref).max() <= 2e-2 * np.abs(ref).max()


def test_forward_is_bidirectional():
    z, cfg, W = _setup()
    ids = z["ids"].copy()
    a = forward(W, cfg, ids, None, emulate_bf16=False)
    ids[:, -1] = (ids[:, -1] + 1) % cfg.vocab_size
    b = forward(W, cfg, ids, None, emulate_bf16=False)
    assert np.abs(a[:, 0] - b[:, 0]).max() > 1e-4
