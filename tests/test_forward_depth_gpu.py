"""The HIP forward at the true LLaDA-8B widths and ALL 32 layers against the numpy oracle (oracle/model.py),
on a small token count so that the oracle finishes in seconds.  The weights are those of the shared full-size
model (LLaDAWeights.random(seed 0)): the same device generator stream is replayed one matrix at a time and each
matrix is handed to the oracle in checkpoint layout, so host memory stays bounded (one layer at a time).

PARITY UNPINNED against the reference (its model code is not in its tree): this pins the HIP forward to the
restatement at full depth.  What is asserted is what two correct bf16 implementations can agree on with random
weights: relative RMS error of the logits, and arg-max agreement on the rows whose top-2 margin is clear of the
noise; the maximum relative error is reported (printed with -s, and kept as a test property)."""
import numpy as np
import pytest

from oracle import model as om

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def replay_weights(cfg, seed=0, std=0.02, head_scale=8.0):
    """Generator of (name, float32 numpy matrix) in the order LLaDAWeights.random draws them."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)

    def w(n, k, s=std):
        return (torch.randn(n, k, generator=g, device="cuda", dtype=torch.float32) * s).to(torch.bfloat16).float().cpu().numpy()

    d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
    yield "wte", w(V, d)
    for i in range(cfg.n_layers):
        for name, shape in (("q_proj", (d, d)), ("k_proj", (d, d)), ("v_proj", (d, d)), ("attn_out", (d, d)),
                            ("ff_proj", (F, d)), ("up_proj", (F, d)), ("ff_out", (d, F))):
            yield f"blocks.{i}.{name}", w(*shape)
    yield "ff_out", w(V, d, std * head_scale)


def test_full_depth_forward_against_the_oracle(model8b, record_property):
    cfg = model8b.cfg
    ocfg = om.LLaDAConfig(d_model=cfg.d_model, n_layers=cfg.n_layers, n_heads=cfg.n_heads, mlp_hidden=cfg.mlp_hidden,
                          vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta, rms_eps=cfg.rms_eps)
    B, L = 2, 64
    g = np.random.Generator(np.random.PCG64(77))
    ids = g.integers(0, 126000, size=(B, L), dtype=np.int64)
    ids[0, 40:] = cfg.mask_token_id                       # a masked tail, as inside generate()
    am = np.ones((B, L), dtype=np.int64)
    am[1, :9] = 0                                         # one left-padded row
    ids[1, :9] = 126081

    got = model8b(torch.from_numpy(ids).cuda(), torch.from_numpy(am).cuda()).logits.float().cpu().numpy()   # [B, L, V]

    stream = replay_weights(cfg)
    name, wte = next(stream)
    assert name == "wte"
    x = wte[ids].astype(np.float32)
    del wte
    cos, sin = om.rope_tables(L, ocfg.head_dim, ocfg.rope_theta)
    ones = np.ones(cfg.d_model, dtype=np.float32)
    for i in range(cfg.n_layers):
        Wl = {"attn_norm": ones, "ff_norm": ones}
        for _ in range(7):
            name, mat = next(stream)
            assert name.startswith(f"blocks.{i}.")
            Wl[name.split(".")[-1]] = mat
        x = om.block(x, Wl, ocfg, cos, sin, am, True)
        del Wl
    name, head = next(stream)
    assert name == "ff_out"
    hf = om.rmsnorm(x, ones, ocfg.rms_eps, True)
    ref = om.round_bf16(hf @ head.T)
    del head

    real = am.astype(bool)                                # logits at padded positions are not defined by the mask
    diff = (got - ref)[real]
    scale = np.abs(ref[real]).max()
    max_rel = float(np.abs(diff).max() / scale)
    rms_rel = float(np.sqrt((diff.astype(np.float64) ** 2).mean()) / np.sqrt((ref[real].astype(np.float64) ** 2).mean()))
    noise = float(np.sqrt((diff.astype(np.float64) ** 2).mean()))
    top2 = np.sort(ref[real], axis=-1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    clear = margin > 4 * noise
    agree_all = float((got[real].argmax(-1) == ref[real].argmax(-1)).mean())
    agree_clear = float((got[real].argmax(-1) == ref[real].argmax(-1))[clear].mean())
    print(f"\nfull depth vs oracle: max_rel {max_rel:.4f}  rms_rel {rms_rel:.5f}  argmax agreement all rows {agree_all:.4f}, "
          f"rows with margin > 4 sigma ({int(clear.sum())} of {clear.size}) {agree_clear:.4f}")
    for k, v in (("max_rel", max_rel), ("rms_rel", rms_rel), ("argmax_all", agree_all), ("argmax_clear", agree_clear)):
        record_property(k, v)
    assert rms_rel <= 1e-2
    assert clear.sum() >= clear.size // 2 and agree_clear >= 0.99
    assert max_rel <= 6e-2        # reported above; two bf16 implementations differ by up to ~4e-2 at 32 layers on random weights
