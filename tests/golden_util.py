"""Helpers shared by the tests: golden-fixture access and the numpy-side fake model."""
import json
import os

import numpy as np

from oracle import rng as orng
from oracle.bf16 import from_bits, round_bf16

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK_ID = 126336


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z["meta_json"]).decode()) if "meta_json" in z.files else None
    return z, meta


def fake_model_logits(seed, call, shape, boost_ids=(), boost=0.0):
    lg = orng.fake_logits(seed, call, shape)
    if boost_ids:
        lg[..., list(boost_ids)] = round_bf16(lg[..., list(boost_ids)] + np.float32(boost))
    return lg


class FakeModel:
    """numpy twin of the fake model the golden tool fed to the reference."""

    def __init__(self, seed, vocab, boost_ids=(), boost=0.0):
        self.seed, self.vocab, self.calls = seed, vocab, 0
        self.boost_ids, self.boost = tuple(boost_ids), boost

    def __call__(self, x, attention_mask=None):
        lg = fake_model_logits(self.seed, self.calls, tuple(x.shape) + (self.vocab,), self.boost_ids, self.boost)
        self.calls += 1
        return lg


class Draws:
    """numpy twin of the RNG replacements of tools/make_golden.py."""

    def __init__(self, noise_seed):
        self.noise_seed, self.n_like, self.n_rand = noise_seed, 0, 0

    def rand_like(self, shape):
        g = np.random.Generator(np.random.PCG64(self.noise_seed + self.n_like))
        self.n_like += 1
        return g.random(tuple(shape))

    def rand(self, shape):
        g = np.random.Generator(np.random.PCG64(self.noise_seed + 7919 + self.n_rand))
        self.n_rand += 1
        return g.random(tuple(shape), dtype=np.float32)


def case_confidence(z, name):
    if f"{name}/confidence_f32" in z.files:
        return z[f"{name}/confidence_f32"]
    return from_bits(z[f"{name}/confidence_bf16"])
