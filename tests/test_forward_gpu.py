"""Parity of the HIP forward and of the end-to-end drop-ins on a small LLaDA-shaped model,
against the oracle (which is pinned to the stand-in vectors, see tests/test_oracle_forward.py)."""
import numpy as np
import pytest

from oracle import likelihood as oll
from oracle import sampler as osamp
from oracle.model import LLaDAConfig as OCfg
from oracle.model import OracleModel, forward, init_weights
from tests.golden_util import load

MASK_ID = 511   # the small test vocabulary (512) needs an in-vocabulary mask token

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def build(ocfg, W, packed=True, tiled=None):
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    cfg = LLaDAConfig(d_model=ocfg.d_model, n_layers=ocfg.n_layers, n_heads=ocfg.n_heads, mlp_hidden=ocfg.mlp_hidden,
                      vocab_size=ocfg.vocab_size, rope_theta=ocfg.rope_theta, rms_eps=ocfg.rms_eps)
    return LLaDAModel(LLaDAWeights.from_numpy(cfg, W, "cuda", packed=packed, tiled=tiled))


@pytest.fixture(scope="module")
def tiny():
    z, _ = load("forward_standin.npz")
    d, nl, nh, ff, V = (int(v) for v in z["cfg"])
    ocfg = OCfg(d_model=d, n_layers=nl, n_heads=nh, mlp_hidden=ff, vocab_size=V)
    W = init_weights(ocfg, seed=int(z["weight_seed"][0]), std=float(z["weight_std"][0]),
                     head_scale=float(z["head_scale"][0]))
    return z, ocfg, W


def max_rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("packed", [True, False])
def test_forward_logits(lib, tiny, packed):
    z, ocfg, W = tiny
    model = build(ocfg, W, packed)
    ids = torch.from_numpy(z["ids"]).cuda()
    for mask, key in ((None, "logits_nomask"), (z["key_mask"], "logits_padmask")):
        am = None if mask is None else torch.from_numpy(mask).cuda()
        lg = model(ids, attention_mask=am).logits
        assert lg.dtype == torch.bfloat16 and tuple(lg.shape) == (2, 24, ocfg.vocab_size)
        got = lg.float().cpu().numpy()
        ref = forward(W, ocfg, z["ids"], mask, emulate_bf16=True)
        assert max_rel(got, ref) <= 2e-2, max_rel(got, ref)        # contract: bf16 tolerance, max rel 2e-2
        assert max_rel(got, z[key]) <= 2e-2, max_rel(got, z[key])  # and against the float32 stand-in itself


@pytest.mark.parametrize("packed", [True, False])
def test_forward_tile_major_weights_are_bit_identical(lib, tiny, packed):
    """Storing the projection matrices tile-major only changes where the GEMMs read them from."""
    z, ocfg, W = tiny
    a, b = build(ocfg, W, packed, tiled=False), build(ocfg, W, packed, tiled=True)
    assert b.weights.tiled and not a.weights.tiled
    ids = torch.from_numpy(z["ids"]).cuda()
    am = torch.from_numpy(z["key_mask"]).cuda()
    assert torch.equal(a(ids, attention_mask=am).logits, b(ids, attention_mask=am).logits)
    rows = torch.tensor([3, 17, 40], dtype=torch.int32, device="cuda")
    assert torch.equal(a.forward_rows(ids, None, rows), b.forward_rows(ids, None, rows))


def test_forward_row_subset_is_exact(lib, tiny):
    z, ocfg, W = tiny
    model = build(ocfg, W)
    ids = torch.from_numpy(z["ids"]).cuda()
    full = model(ids).logits.reshape(-1, ocfg.vocab_size)
    rows = torch.tensor([47, 0, 5, 23, 24, 31], dtype=torch.int32, device="cuda")
    sub = model.forward_rows(ids, None, rows)
    assert torch.equal(sub, full[rows.long()])


def test_true_shape_layer(lib):
    """One layer at the real LLaDA-8B widths (d 4096, 32 heads, FFN 12288), ragged sequence length."""
    ocfg = OCfg(d_model=4096, n_layers=1, n_heads=32, mlp_hidden=12288, vocab_size=1024)
    W = init_weights(ocfg, seed=3, std=0.02, head_scale=8.0)
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(1))
    ids = g.integers(0, 1024, size=(2, 75), dtype=np.int64)
    km = np.ones((2, 75), dtype=np.int64)
    km[1, :9] = 0
    got = model(torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(km).cuda()).logits.float().cpu().numpy()
    ref = forward(W, ocfg, ids, km, emulate_bf16=True)
    assert max_rel(got, ref) <= 2e-2, max_rel(got, ref)


def test_true_shape_last_layer_rows_only(lib):
    """With few wanted rows the last layer's output projection and MLP run on those rows alone
    (small-M GEMMs, split-K): same logits as the all-rows forward up to accumulation order."""
    from oracle.bf16 import ulp_distance
    ocfg = OCfg(d_model=4096, n_layers=2, n_heads=32, mlp_hidden=12288, vocab_size=1024)
    W = init_weights(ocfg, seed=4, std=0.02, head_scale=8.0)
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(2))
    ids = torch.from_numpy(g.integers(0, 1024, size=(2, 96), dtype=np.int64)).cuda()
    full = model(ids).logits.reshape(-1, 1024).float().cpu().numpy()
    rows = np.concatenate([np.arange(40, 72), np.arange(96 + 40, 96 + 72)]).astype(np.int32)
    sub = model.forward_rows(ids, None, torch.from_numpy(rows).cuda()).float().cpu().numpy()
    ref = full[rows]
    assert max_rel(sub, ref) <= 2.0 ** -7, max_rel(sub, ref)      # one bf16 step of the largest logit
    assert (ulp_distance(sub, ref) <= 1).mean() >= 0.95            # small logits move by more ulps


def test_generate_end_to_end(lib, tiny):
    """llada_amd.generate on the HIP model (fast path: LM head on the block rows only), batch with
    left padding.  Every step is checked teacher-forced: the oracle's confidence_step on the HIP
    model's own full logits for that step's input must give the same candidates/confidences, and the
    tokens that were transferred must be a valid top-k of them (bit-exact sampler parity given
    identical logits; logits parity itself is test_forward_logits).  The free-running comparison
    against the oracle model is informative only: a random 512-word model has near-tied logits, so a
    1-ulp logit difference flips tokens and the flip propagates."""
    from llada_amd import generate
    from oracle.bf16 import ulp_distance
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(12))
    B, P, G = 3, 10, 32
    prompt = g.integers(0, MASK_ID, size=(B, P), dtype=np.int64)
    am = np.ones((B, P), dtype=np.int64)
    am[1, :3] = 0
    om = OracleModel(W, ocfg, True)
    same = total = 0
    for (steps, bl, mask) in ((32, 8, am), (8, 32, None), (16, 16, am)):
        trace = []

        def cb(block, step, x, x0, conf):
            trace.append((block, x.cpu().numpy().copy(), x0.cpu().numpy().copy(), conf.cpu().numpy().copy()))

        mt = None if mask is None else torch.from_numpy(mask)
        out = generate(model, torch.from_numpy(prompt), mt, steps=steps, gen_length=G, block_length=bl, temperature=0.,
                       cfg_scale=0., remasking="low_confidence", mask_id=MASK_ID, step_callback=cb).cpu().numpy()
        assert np.array_equal(out[:, :P], prompt) and not (out == MASK_ID).any()
        full_mask = None if mask is None else torch.from_numpy(np.concatenate([mask, np.ones((B, G), np.int64)], 1)).cuda()
        x_in = np.full((B, P + G), MASK_ID, dtype=np.int64)
        x_in[:, :P] = prompt
        spb = steps // (G // bl)
        for t, (block, x_after, x0, conf) in enumerate(trace):
            lg = model(torch.from_numpy(x_in).cuda(), attention_mask=full_mask).logits.float().cpu().numpy()
            ox0, oconf = osamp.confidence_step(x_in, lg, MASK_ID, P + (block + 1) * bl, "low_confidence")
            live = ~np.isneginf(oconf)
            assert np.array_equal(np.isneginf(conf), np.isneginf(oconf))
            assert np.array_equal(x0[live], ox0[live])
            assert ulp_distance(conf[live], oconf[live]).max(initial=0) <= 1
            changed = x_after != x_in
            k = osamp.get_num_transfer_tokens(x_in[:, P + block * bl: P + (block + 1) * bl] == MASK_ID, spb - t % spb)
            for j in range(B):
                assert osamp.is_valid_topk(conf[j], changed[j], changed[j].sum())
            assert np.array_equal(x_after[changed], x0[changed])
            x_in = x_after
        assert np.array_equal(x_in, out)
        ref = osamp.generate(om, prompt, mask, steps=steps, gen_length=G, block_length=bl, temperature=0.0,
                             cfg_scale=0.0, remasking="low_confidence", mask_id=MASK_ID)
        same += int((out[:, P:] == ref[:, P:]).sum())
        total += B * G
    print(f"free-running agreement with the oracle model: {same}/{total}")
    assert same >= 0.5 * total, (same, total)


def test_generate_cfg_gumbel_end_to_end(lib, tiny):
    """CFG + temperature with injected noise: per-step teacher-forced agreement of the candidate
    tokens on the positions both sides transfer (free-running sampling amplifies 1-ulp logit
    differences, so the end-to-end check is per step)."""
    from llada_amd import generate
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(13))
    B, P, G, bl, steps = 2, 6, 16, 8, 8
    prompt = g.integers(0, MASK_ID, size=(B, P), dtype=np.int64)
    am = np.ones((B, P), dtype=np.int64)
    seeds = iter(range(1000, 2000))
    noise_log = []

    def noise_fn(shape):
        u = np.random.Generator(np.random.PCG64(next(seeds))).random(shape)
        noise_log.append(u)
        return torch.from_numpy(u)

    out = generate(model, torch.from_numpy(prompt), torch.from_numpy(am), steps=steps, gen_length=G, block_length=bl,
                   temperature=0.7, cfg_scale=1.0, remasking="low_confidence", mask_id=MASK_ID, noise_fn=noise_fn)
    out = out.cpu().numpy()
    assert not (out == MASK_ID).any() and np.array_equal(out[:, :P], prompt)
    it = iter(noise_log)
    ref = osamp.generate(OracleModel(W, ocfg, True), prompt, am, steps=steps, gen_length=G, block_length=bl,
                         temperature=0.7, cfg_scale=1.0, remasking="low_confidence", mask_id=MASK_ID,
                         noise_fn=lambda s: next(it))
    assert (out == ref).mean() >= 0.9


def test_likelihood_end_to_end(lib, tiny):
    """llada_amd.get_log_likelihood on the HIP model vs the oracle on the oracle model with the same
    masks: within 1e-3 relative (the contract)."""
    from llada_amd import get_log_likelihood
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(14))
    prompt = g.integers(0, MASK_ID, size=9, dtype=np.int64)
    answer = g.integers(0, MASK_ID, size=15, dtype=np.int64)
    for cfg_scale in (0.0, 0.5):
        dg = np.random.Generator(np.random.PCG64(15))
        ks = [int(dg.integers(1, 16)) for _ in range(4)]
        perms = [dg.permutation(15) for _ in range(4 * 8)]
        k1, p1 = list(ks), iter(perms)
        val = get_log_likelihood(model, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=32, batch_size=8,
                                 cfg_scale=cfg_scale, mask_id=MASK_ID, k_fn=lambda T: k1.pop(0),
                                 perm_fn=lambda T: next(p1), round_ce_bf16=False)
        k2, p2 = list(ks), iter(perms)
        ref = oll.get_log_likelihood(OracleModel(W, ocfg, True), prompt, answer, mc_num=32, batch_size=8,
                                     cfg_scale=cfg_scale, mask_id=MASK_ID, k_fn=lambda T: k2.pop(0),
                                     perm_fn=lambda T: next(p2), round_to_bf16=False)
        assert abs(val - ref) <= 1e-3 * abs(ref), (val, ref)
    # library-RNG path runs and is reproducible
    a = get_log_likelihood(model, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=16, batch_size=8, mask_id=MASK_ID, seed=5)
    b = get_log_likelihood(model, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=16, batch_size=8, mask_id=MASK_ID, seed=5)
    assert a == b and a < 0


def test_generate_mask_tokens_left_of_the_block(lib, tiny):
    """generate.py:125-128 excludes only the positions AFTER the current block: a mask token inside the prompt
    (or one the model predicted in an earlier block) is a transfer candidate in every block.  The fast path then
    asks the forward for those rows as well; every step is checked against the oracle's epilogue on the model's
    own full logits, and the transfers against the schedule of generate.py:75-79."""
    from llada_amd import generate
    from oracle.bf16 import ulp_distance
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(21))
    B, P, G, bl, steps = 2, 12, 16, 8, 8
    prompt = g.integers(0, MASK_ID, size=(B, P), dtype=np.int64)
    prompt[0, 3] = MASK_ID
    prompt[1, [0, 7]] = MASK_ID
    trace = []
    out = generate(model, torch.from_numpy(prompt), steps=steps, gen_length=G, block_length=bl, temperature=0.,
                   remasking="low_confidence", mask_id=MASK_ID,
                   step_callback=lambda b, s, x, x0, c: trace.append((b, x.cpu().numpy().copy(), x0.cpu().numpy().copy(),
                                                                      c.cpu().numpy().copy()))).cpu().numpy()
    x_in = np.full((B, P + G), MASK_ID, dtype=np.int64)
    x_in[:, :P] = prompt
    spb = steps // (G // bl)
    prompt_candidates = 0
    for t, (block, x_after, x0, conf) in enumerate(trace):
        lg = model(torch.from_numpy(x_in).cuda()).logits.float().cpu().numpy()
        ox0, oconf = osamp.confidence_step(x_in, lg, MASK_ID, P + (block + 1) * bl, "low_confidence")
        live = ~np.isneginf(oconf)
        prompt_candidates += int(live[:, :P].sum())
        assert np.array_equal(np.isneginf(conf), np.isneginf(oconf))
        assert np.array_equal(x0[live], ox0[live])
        assert ulp_distance(conf[live], oconf[live]).max(initial=0) <= 1
        if t % spb == 0:
            sched = osamp.get_num_transfer_tokens(x_in[:, P + block * bl: P + (block + 1) * bl] == MASK_ID, spb)
        expect = x_in.copy()
        osamp.topk_transfer(expect, x0, conf, sched[:, t % spb])
        assert np.array_equal(x_after, expect)
        x_in = x_after
    assert prompt_candidates > 0            # the case really occurred
    assert np.array_equal(x_in, out)


@pytest.mark.parametrize("mc_num,batch_size,cfg_scale", [(1, 1, 0.0), (4, 2, 0.0), (6, 3, 0.5), (8, 8, 0.0)])
def test_likelihood_many_matches_one_by_one(lib, tiny, mc_num, batch_size, cfg_scale):
    """Requests of different lengths batched into shared, right-padded forwards give the values of the
    one-by-one calls with the same seeds (same masks; only the GEMM tiling of the forward differs)."""
    from llada_amd import get_log_likelihood, get_log_likelihood_many
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(31))
    pairs = []
    for l1, T in ((9, 15), (0, 4), (30, 1), (17, 40), (5, 5), (12, 33), (3, 70)):
        pairs.append((torch.from_numpy(g.integers(0, MASK_ID, size=l1, dtype=np.int64)),
                      torch.from_numpy(g.integers(0, MASK_ID, size=T, dtype=np.int64))))
    seeds = [100 + 7 * i for i in range(len(pairs))]
    many = get_log_likelihood_many(model, pairs, mc_num=mc_num, batch_size=batch_size, cfg_scale=cfg_scale, mask_id=MASK_ID,
                                   seeds=seeds, max_rows=10)
    for (p, a), s, v in zip(pairs, seeds, many):
        one = get_log_likelihood(model, p, a, mc_num=mc_num, batch_size=batch_size, cfg_scale=cfg_scale, mask_id=MASK_ID,
                                 seed=s)
        assert abs(v - one) <= 1e-3 * abs(one), (len(p), len(a), v, one)


@pytest.mark.parametrize("packed", [True, False])
def test_forward_with_explicit_positions_both_weight_layouts(lib, tiny, packed):
    """Explicit rotary positions in the fused QKV epilogue (packed weights) and in the separate rotary kernel
    (checkpoint layout): an arbitrary position table gives the oracle's logits for the same table -- the oracle
    rotates with the table rows picked by position -- and both layouts give the same logits."""
    from oracle import model as om
    z, ocfg, W = tiny
    model = build(ocfg, W, packed)
    ids = z["ids"]
    B, L = ids.shape
    g = np.random.Generator(np.random.PCG64(5))
    pos = np.stack([g.permutation(L), np.minimum(np.arange(L) // 2 + 3, L - 1)]).astype(np.int32)   # any rows of the table
    got = model(torch.from_numpy(ids).cuda(), position_ids=torch.from_numpy(pos).cuda()).logits.float().cpu().numpy()

    # oracle with the same per-token table rows
    cos, sin = om.rope_tables(L, ocfg.head_dim, ocfg.rope_theta)
    x = W["wte"][ids].astype(np.float32)
    H, D = ocfg.n_heads, ocfg.head_dim
    for i in range(ocfg.n_layers):
        Wl = {k: W[f"blocks.{i}.{k}"] for k in om.LAYER_KEYS}
        h = om.rmsnorm(x, Wl["attn_norm"], ocfg.rms_eps, True)
        q, k, v = (om.round_bf16(h @ Wl[n].T).reshape(B, L, H, D).transpose(0, 2, 1, 3) for n in ("q_proj", "k_proj", "v_proj"))
        qr, kr = np.empty_like(q), np.empty_like(k)
        for b in range(B):
            qr[b:b + 1] = om.apply_rope(q[b:b + 1], cos[pos[b]], sin[pos[b]], True)
            kr[b:b + 1] = om.apply_rope(k[b:b + 1], cos[pos[b]], sin[pos[b]], True)
        a = om.attention(qr, kr, v, None, True).transpose(0, 2, 1, 3).reshape(B, L, H * D)
        x = om.round_bf16(x + om.round_bf16(a @ Wl["attn_out"].T))
        h = om.rmsnorm(x, Wl["ff_norm"], ocfg.rms_eps, True)
        gt, up = om.round_bf16(h @ Wl["ff_proj"].T), om.round_bf16(h @ Wl["up_proj"].T)
        act = om.round_bf16(gt / (np.float32(1.0) + np.exp(-gt)))
        x = om.round_bf16(x + om.round_bf16(om.round_bf16(act * up) @ Wl["ff_out"].T))
    ref = om.round_bf16(om.rmsnorm(x, W["ln_f"], ocfg.rms_eps, True) @ W["ff_out"].T)
    assert max_rel(got, ref) <= 2e-2, max_rel(got, ref)
    plain = model(torch.from_numpy(ids).cuda()).logits.float().cpu().numpy()
    assert max_rel(plain, ref) > 5e-2            # the table rows matter: column numbering is another function here


def test_generate_with_guidance_and_positions_from_mask(lib, tiny):
    """generate(position_ids="from_mask") under classifier-free guidance (the conditional and the unconditional copy
    share the positions) on a left-padded batch: every step's candidates are those of the same call without explicit
    positions wherever the top-2 margin is clear (both numberings are the same function up to rounding), and the
    transfer counts are exact."""
    from llada_amd import generate
    z, ocfg, W = tiny
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(9))
    B, P, G, bl = 3, 12, 16, 8
    prompt = torch.from_numpy(g.integers(0, MASK_ID, size=(B, P), dtype=np.int64))
    am = torch.ones(B, P, dtype=torch.long)
    am[1, :5] = 0
    am[2, :9] = 0
    counts = []
    out = generate(model, prompt, am, steps=8, gen_length=G, block_length=bl, temperature=0., cfg_scale=1.5, mask_id=MASK_ID,
                   position_ids="from_mask", step_callback=lambda b, s, x, x0, c: counts.append(int((x == MASK_ID).sum())))
    assert counts == [B * (G - 2 * (i + 1)) for i in range(8)] and not (out == MASK_ID).any()
    with pytest.raises(AssertionError):
        generate(model, prompt, None, steps=8, gen_length=G, block_length=bl, mask_id=MASK_ID, position_ids="from_mask")
    with pytest.raises(NotImplementedError):
        generate(model, prompt, am, steps=8, gen_length=G, block_length=bl, mask_id=MASK_ID, position_ids="from_mask", cache="dual")
