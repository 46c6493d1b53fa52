"""The HIP forward at the true LLaDA-8B widths and ALL 32 layers against the numpy oracle (oracle/model.py),
on a small token count so that the oracle finishes in seconds.  The weights are those LLaDAWeights.random draws:
the same device generator stream is replayed one matrix at a time and each matrix is handed to the oracle in
checkpoint layout, so host memory stays bounded (one layer at a time).

PARITY UNPINNED against the reference (its model code is not in its tree): this pins the HIP forward to the
restatement at full depth, in two ways.

  * Layer by layer (teacher-forced): the oracle's block is applied to the HIP forward's OWN residual stream after l
    layers and must reproduce the HIP stream after l + 1 layers -- for every one of the 32 layers, with that layer's
    own weights -- and the oracle's final norm + LM head on the HIP stream after the last layer must reproduce the
    HIP logits.  Nothing accumulates across layers here, so the contract bounds are asserted as written (logits
    max-rel <= 2e-2, arg-max agreement >= 99 %), plus a bound on every layer's update that a kernel error of a
    few 1e-2 in any projection, the attention or a norm could not pass.
  * Free-running over all 32 layers: with random weights every bf16 rounding is amplified from layer to layer, so
    the oracle is run twice -- with the roundings of a bf16 model run, and in plain float32 -- and what is asserted
    is that the HIP logits are as close to the float32 arithmetic as the bf16-emulating oracle's are (the noise
    floor), plus arg-max agreement on the rows whose top-2 margin is clear of the noise.

Both are run on the i.i.d. N(0, 0.02^2) weights of the shared full-size model and on a depth-scaled initialisation
(the two residual writers attn_out / ff_out at 0.02 / sqrt(2 n_layers)).  The relative errors are reported (printed
with -s, and kept as test properties)."""
import numpy as np
import pytest

from oracle import model as om

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def replay_weights(cfg, seed=0, std=0.02, head_scale=8.0, resid_std=None, embed_std=None):
    """Generator of (name, float32 numpy matrix) in the order LLaDAWeights.random draws them."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    rs = std if resid_std is None else resid_std

    def w(n, k, s=std):
        return (torch.randn(n, k, generator=g, device="cuda", dtype=torch.float32) * s).to(torch.bfloat16).float().cpu().numpy()

    d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
    yield "wte", w(V, d, std if embed_std is None else embed_std)
    for i in range(cfg.n_layers):
        for name, shape, s in (("q_proj", (d, d), std), ("k_proj", (d, d), std), ("v_proj", (d, d), std),
                               ("attn_out", (d, d), rs), ("ff_proj", (F, d), std), ("up_proj", (F, d), std),
                               ("ff_out", (d, F), rs)):
            yield f"blocks.{i}.{name}", w(*shape, s)
    yield "ff_out", w(V, d, std * head_scale)


def rms(a):
    return float(np.sqrt((a.astype(np.float64) ** 2).mean()))


def agree(a, b):
    return float((a.argmax(-1) == b.argmax(-1)).mean())


def run_depth(model, resid_std, record_property, tag, embed_std=None):
    cfg = model.cfg
    ocfg = om.LLaDAConfig(d_model=cfg.d_model, n_layers=cfg.n_layers, n_heads=cfg.n_heads, mlp_hidden=cfg.mlp_hidden,
                          vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta, rms_eps=cfg.rms_eps)
    B, L = 2, 64
    g = np.random.Generator(np.random.PCG64(77))
    ids = g.integers(0, 126000, size=(B, L), dtype=np.int64)
    ids[0, 40:] = cfg.mask_token_id                       # a masked tail, as inside generate()
    am = np.ones((B, L), dtype=np.int64)
    am[1, :9] = 0                                         # one left-padded row
    ids[1, :9] = 126081
    real = am.astype(bool)                                # positions behind the padding mask are not defined by it
    ids_t, am_t = torch.from_numpy(ids).cuda(), torch.from_numpy(am).cuda()

    got = model(ids_t, am_t).logits.float().cpu().numpy()   # [B, L, V]
    # the HIP forward's own residual stream after 0 .. n_layers blocks
    hip_x = [model.residual_stream(ids_t, am_t, l).float().cpu().numpy() for l in range(cfg.n_layers + 1)]

    # oracle runs over the same weights: free-running with the bf16 roundings of a bf16 model run (what the HIP path
    # is a faithful implementation of), free-running in plain float32 (the exact arithmetic both approximate), and
    # one block at a time on the HIP stream (teacher-forced)
    stream = replay_weights(cfg, resid_std=resid_std, embed_std=embed_std)
    name, wte = next(stream)
    assert name == "wte"
    xb = wte[ids].astype(np.float32)
    xf = xb.copy()
    assert np.array_equal(hip_x[0], xb)                   # the embedding lookup is exact
    del wte
    cos, sin = om.rope_tables(L, ocfg.head_dim, ocfg.rope_theta)
    ones = np.ones(cfg.d_model, dtype=np.float32)
    layer_max, layer_upd = [], []
    for i in range(cfg.n_layers):
        Wl = {"attn_norm": ones, "ff_norm": ones}
        for _ in range(7):
            name, mat = next(stream)
            assert name.startswith(f"blocks.{i}.")
            Wl[name.split(".")[-1]] = mat
        xb = om.block(xb, Wl, ocfg, cos, sin, am, True)
        xf = om.block(xf, Wl, ocfg, cos, sin, am, False)
        tf = om.block(hip_x[i], Wl, ocfg, cos, sin, am, True)                 # the oracle's layer on the HIP input
        d_ = (hip_x[i + 1] - tf)[real]
        layer_max.append(float(np.abs(d_).max() / np.abs(tf[real]).max()))
        layer_upd.append(rms(d_) / rms((tf - hip_x[i])[real]))                # relative to what the layer added
        del Wl
    name, head = next(stream)
    assert name == "ff_out"
    ref = om.round_bf16(om.rmsnorm(xb, ones, ocfg.rms_eps, True) @ head.T)
    exact = om.rmsnorm(xf, ones, ocfg.rms_eps, False) @ head.T
    ref_tf = om.round_bf16(om.rmsnorm(hip_x[-1], ones, ocfg.rms_eps, True) @ head.T)
    del head

    got, ref, exact, ref_tf = got[real], ref[real], exact[real], ref_tf[real]
    # ---- teacher-forced: the contract bounds as written ----
    tf_max_rel = float(np.abs(got - ref_tf).max() / np.abs(ref_tf).max())
    tf_agree = agree(got, ref_tf)
    print(f"\n[{tag}] layer by layer on the HIP stream ({cfg.n_layers} layers, d {cfg.d_model}, {int(real.sum())} tokens): stream "
          f"max-rel worst {max(layer_max):.2e} (layer {int(np.argmax(layer_max))}), update rms-rel worst {max(layer_upd):.2e} "
          f"(layer {int(np.argmax(layer_upd))}), median {np.median(layer_upd):.2e}; final norm + LM head on the HIP stream: logits "
          f"max-rel {tf_max_rel:.2e}, arg-max agreement {tf_agree:.4f}")
    # ---- free-running over all layers ----
    scale = np.abs(ref).max()
    max_rel = float(np.abs(got - ref).max() / scale)
    e_pair = rms(got - ref) / rms(ref)                    # HIP vs the bf16-emulating oracle
    e_hip = rms(got - exact) / rms(exact)                 # HIP vs exact arithmetic
    e_orc = rms(ref - exact) / rms(exact)                 # bf16-emulating oracle vs exact arithmetic: the noise floor
    noise = rms(got - ref)
    top2 = np.sort(ref, axis=-1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 4 * noise
    a_pair, a_hip, a_orc = agree(got, ref), agree(got, exact), agree(ref, exact)
    a_clear = float((got.argmax(-1) == ref.argmax(-1))[clear].mean()) if clear.any() else float("nan")
    print(f"[{tag}] free-running: HIP vs bf16 oracle max_rel {max_rel:.4f} rms_rel {e_pair:.4f} | vs float32 arithmetic: HIP "
          f"{e_hip:.4f}, bf16 oracle {e_orc:.4f} | arg-max agreement: HIP/oracle {a_pair:.3f}, HIP/float32 {a_hip:.3f}, "
          f"oracle/float32 {a_orc:.3f}, HIP/oracle on the {int(clear.sum())} rows with top-2 margin > 4 sigma {a_clear:.3f}")
    res = dict(layer_stream_max_rel=max(layer_max), layer_update_rms_rel=max(layer_upd), tf_logits_max_rel=tf_max_rel,
               tf_argmax=tf_agree, max_rel=max_rel, rms_rel_pair=e_pair, rms_rel_hip_vs_f32=e_hip, rms_rel_oracle_vs_f32=e_orc,
               argmax_pair=a_pair, argmax_hip_f32=a_hip, argmax_oracle_f32=a_orc, argmax_clear=a_clear, n_clear=int(clear.sum()))
    for k, v in res.items():
        record_property(f"{tag}_{k}", v)
    return res


def check_layerwise(res):
    """Contract bounds as written, on every layer in isolation: a kernel error cannot hide behind the noise that
    random weights accumulate over depth."""
    assert res["layer_stream_max_rel"] <= 2e-2
    assert res["layer_update_rms_rel"] <= 2e-2            # a wrong projection / attention / norm shows up at order 1e-1 .. 1
    assert res["tf_logits_max_rel"] <= 2e-2
    assert res["tf_argmax"] >= 0.99


def check_free_running(res):
    """The HIP forward is as close to exact arithmetic as a faithful bf16 run is (random weights amplify every
    rounding over 32 layers, so two correct bf16 runs differ from each other by about sqrt(2) of that noise)."""
    assert res["rms_rel_hip_vs_f32"] <= 1.25 * res["rms_rel_oracle_vs_f32"] + 1e-3
    assert res["rms_rel_pair"] <= 1.6 * max(res["rms_rel_hip_vs_f32"], res["rms_rel_oracle_vs_f32"])
    assert res["argmax_hip_f32"] >= res["argmax_oracle_f32"] - 0.08
    assert res["n_clear"] >= 16 and res["argmax_clear"] >= 0.99


def test_full_depth_forward_against_the_oracle(model8b, record_property):
    res = run_depth(model8b, None, record_property, "iid")
    check_layerwise(res)
    check_free_running(res)


def test_full_depth_forward_depth_scaled_init(lib, record_property):
    """The same on a depth-scaled initialisation: residual writers at std 0.02 / sqrt(2 * 32), everything else 0.02.
    Measured: the free-running figures do not move (5.2e-2 RMS against 5.1e-2 for the i.i.d. weights).  A pre-norm
    stack is invariant to a common scale of everything written into the residual stream: the branches see the
    normalised stream, so scaling both writers scales the whole stream (the 0.02 embedding is negligible next to the
    first layer's update either way) and every layer still moves it by the same relative amount."""
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    cfg = LLaDAConfig()
    rs = 0.02 / np.sqrt(2.0 * cfg.n_layers)
    model = LLaDAModel(LLaDAWeights.random(cfg, "cuda", seed=0, resid_std=rs))
    try:
        res = run_depth(model, rs, record_property, "depth_scaled")
    finally:
        del model
        torch.cuda.empty_cache()
    check_layerwise(res)
    check_free_running(res)


def test_full_depth_forward_embedding_dominated_stream(lib, record_property):
    """What conditions a pre-norm stack is the ratio embedding : layer update.  Embedding at std 4 with the depth-scaled
    writers: every layer changes the stream by a few per cent instead of replacing it, and a rounding made in one
    layer is hardly amplified by the layers above.  Same 32 layers, same widths, same kernels, same per-layer errors
    -- and the free-running comparison over all 32 layers is then asserted against the contract's logits bound as
    written (max-rel <= 2e-2 against the bf16 oracle).  Measured on the way: embedding std 0.02 (i.i.d. weights)
    5.5e-2, std 1 2.6e-2 (arg-max agreement 0.84 -> 0.96): the free-running distance follows the conditioning of the
    weights, the per-layer errors do not move."""
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    cfg = LLaDAConfig()
    rs = 0.02 / np.sqrt(2.0 * cfg.n_layers)
    model = LLaDAModel(LLaDAWeights.random(cfg, "cuda", seed=0, resid_std=rs, embed_std=4.0))
    try:
        res = run_depth(model, rs, record_property, "embedding_dominated", embed_std=4.0)
    finally:
        del model
        torch.cuda.empty_cache()
    check_layerwise(res)
    assert res["max_rel"] <= 2e-2                                  # measured 1.1e-2
    assert res["rms_rel_hip_vs_f32"] <= 1.25 * res["rms_rel_oracle_vs_f32"] + 1e-3
    assert res["argmax_clear"] >= 0.98                             # measured 0.991 (one of 106 clear-margin rows)
