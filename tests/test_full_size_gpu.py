"""The drop-ins at the full LLaDA-8B shapes (32 layers, d 4096, FFN 12288, vocabulary 126464, random
weights), where the numpy oracle cannot run the model: size-independent properties of the sampler
(exact transfer counts, monotone unmasking inside the current block, determinism), the epilogue
and the masked cross-entropy against the oracle on the HIP forward's own logits."""
import numpy as np
import pytest

from oracle import likelihood as oll
from oracle import sampler as osamp
from oracle.bf16 import ulp_distance

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MASK_ID = 126336


def prompts(B, P, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 126000, (B, P), generator=g, dtype=torch.long)


def test_generate_single_sequence_invariants(model8b):
    """Batch 1, gen_length 256, steps 256, block_length 32 (one token per step): every step reveals
    exactly one position, inside the current block, and never touches a revealed one; two runs
    are bit-identical."""
    from llada_amd import generate
    P, G, bl = 128, 256, 32
    prompt = prompts(1, P)
    trace = []
    out = generate(model8b, prompt, steps=G, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence",
                   step_callback=lambda b, s, x, x0, c: trace.append((b, x.clone())))
    assert len(trace) == G
    xs = torch.stack([t[1] for t in trace]).cpu().numpy()[:, 0]                     # [steps, L]
    prev = np.concatenate([prompt[0].numpy(), np.full(G, MASK_ID)])
    for n, (block, _) in enumerate(trace):
        cur = xs[n]
        changed = np.flatnonzero(cur != prev)
        assert len(changed) == 1, (n, changed)
        assert prev[changed[0]] == MASK_ID and cur[changed[0]] != MASK_ID
        assert P + block * bl <= changed[0] < P + (block + 1) * bl
        prev = cur
    o = out.cpu().numpy()
    assert np.array_equal(o[0], prev) and np.array_equal(o[0, :P], prompt[0].numpy()) and not (o == MASK_ID).any()
    again = generate(model8b, prompt, steps=G, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence")
    assert torch.equal(out, again)


def test_generate_reference_cpu_configuration_shape(model8b):
    """The configuration the reference's CPU path is quoted on, on the HIP path: batch 1, P = 128, gen_length 64,
    steps 64, block_length 32, temperature 0, low_confidence (generate.py:74-134).  Every one of the 64 steps
    reveals exactly one token inside its block (two blocks of 32), a revealed token never changes, two runs are
    bit-identical, and on two steps (the first of each block) the oracle's epilogue on the HIP forward's own
    full-vocabulary logits of the step's input gives the same candidates, confidences and transferred token."""
    from llada_amd import generate
    B, P, G, bl = 1, 128, 64, 32
    L = P + G
    prompt = prompts(B, P, seed=64)
    trace = []
    out = generate(model8b, prompt, steps=G, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence",
                   step_callback=lambda b, s, x, x0, c: trace.append((b, s, x.cpu().numpy().copy(), x0.cpu().numpy().copy(),
                                                                      c.cpu().numpy().copy())))
    assert [(t[0], t[1]) for t in trace] == [(n // bl, n % bl) for n in range(G)]
    start = np.concatenate([prompt[0].numpy(), np.full(G, MASK_ID)])[None, :]
    prev = start
    for n, (block, _, x_after, _, _) in enumerate(trace):
        changed = np.flatnonzero(x_after[0] != prev[0])
        assert len(changed) == 1, (n, changed)                                       # get_num_transfer_tokens: 32 / 32 = 1
        assert prev[0, changed[0]] == MASK_ID and x_after[0, changed[0]] != MASK_ID
        assert P + block * bl <= changed[0] < P + (block + 1) * bl                   # :125 confines it to the block
        assert (x_after[0, P + (block + 1) * bl:] == MASK_ID).all()
        prev = x_after
    o = out.cpu().numpy()
    assert np.array_equal(o, prev) and np.array_equal(o[0, :P], prompt[0].numpy()) and not (o == MASK_ID).any()
    again = generate(model8b, prompt, steps=G, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence")
    assert torch.equal(out, again)
    for n in (0, bl):                                                                # first step of each block
        x_in = start if n == 0 else trace[n - 1][2]
        block = n // bl
        # logits of the block's positions (every other position's are discarded by generate.py:125-128), asked
        # for the way generate() asks, so the GEMM plan and hence every rounding is the step's own
        rows = torch.arange(P + block * bl, P + (block + 1) * bl, dtype=torch.int32).cuda().contiguous()
        lg = model8b.forward_rows(torch.from_numpy(x_in).cuda(), None, rows).float().cpu().numpy()
        full = np.zeros((1, L, lg.shape[-1]), dtype=np.float32)
        full[0, P + block * bl:P + (block + 1) * bl] = lg
        ox0, oconf = osamp.confidence_step(x_in, full, MASK_ID, P + (block + 1) * bl, "low_confidence")
        _, _, x_after, x0, conf = trace[n]
        live = ~np.isneginf(oconf)
        assert live.sum() == bl and np.array_equal(live, ~np.isneginf(conf))
        assert np.array_equal(x0[live], ox0[live])
        assert ((ulp_distance(conf[live], oconf[live]) <= 1) | (np.abs(conf[live] - oconf[live]) <= 2e-2)).all()
        x_next = x_in.copy()
        osamp.topk_transfer(x_next, x0, conf, np.ones(1, dtype=np.int64))            # :130-134 on the step's own scores
        assert np.array_equal(x_next, x_after)


def test_generate_batched_padded_schedule(model8b):
    """8 rows (one left-padded with an attention mask), 4 tokens per step: every row reveals exactly
    get_num_transfer_tokens' count per step; sampled runs (temperature, library noise) are
    reproducible for a seed and differ between seeds."""
    from llada_amd import generate
    B, P, G, bl, steps = 8, 128, 64, 32, 16
    prompt = prompts(B, P, seed=7)
    am = torch.ones(B, P, dtype=torch.long)
    am[3, :17] = 0
    prompt[3, :17] = 126081
    trace = []
    out = generate(model8b, prompt, am, steps=steps, gen_length=G, block_length=bl, temperature=0.,
                   remasking="low_confidence", step_callback=lambda b, s, x, x0, c: trace.append(x.clone()))
    xs = torch.stack(trace).cpu().numpy()
    masked = (xs == MASK_ID).sum(axis=2)                                             # [steps, B]
    assert np.array_equal(masked, np.repeat((G - 4 * np.arange(1, steps + 1))[:, None], B, axis=1))
    assert not (out == MASK_ID).any()
    a = generate(model8b, prompt[:2], am[:2], steps=steps, gen_length=G, block_length=bl, temperature=0.7,
                 cfg_scale=0.5, remasking="low_confidence", seed=5)
    b = generate(model8b, prompt[:2], am[:2], steps=steps, gen_length=G, block_length=bl, temperature=0.7,
                 cfg_scale=0.5, remasking="low_confidence", seed=5)
    c = generate(model8b, prompt[:2], am[:2], steps=steps, gen_length=G, block_length=bl, temperature=0.7,
                 cfg_scale=0.5, remasking="low_confidence", seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("remasking", ["low_confidence", "high_entropy"])
def test_step_epilogue_matches_oracle_on_hip_logits(model8b, remasking):
    """One denoising step at the full vocabulary: the oracle's epilogue on the logits the HIP forward
    produced for the block rows gives the same candidates (exact) and confidences (1 bf16 ulp)."""
    from llada_amd import generate
    B, P, G, bl = 2, 128, 64, 32
    prompt = prompts(B, P, seed=11)
    got = []
    generate(model8b, prompt, steps=2, gen_length=G, block_length=bl, temperature=0., remasking=remasking,
             step_callback=lambda b, s, x, x0, c: got.append((x0.clone(), c.clone())))
    L = P + G
    x_in = torch.full((B, L), MASK_ID, dtype=torch.long)
    x_in[:, :P] = prompt
    rows = (torch.arange(B, dtype=torch.int32)[:, None] * L + torch.arange(P, P + bl, dtype=torch.int32)[None, :]).reshape(-1)
    lg = model8b.forward_rows(x_in.cuda(), None, rows.cuda().contiguous()).float().cpu().numpy()     # [B*bl, V]
    full = np.zeros((B, L, lg.shape[-1]), dtype=np.float32)
    full[:, P:P + bl] = lg.reshape(B, bl, -1)
    ox0, oconf = osamp.confidence_step(x_in.numpy(), full, MASK_ID, P + bl, remasking)
    x0, conf = got[0][0].cpu().numpy(), got[0][1].cpu().numpy()
    live = ~np.isneginf(oconf)
    assert live.sum() == B * bl and np.array_equal(np.isneginf(conf), np.isneginf(oconf))
    assert np.array_equal(x0[live], ox0[live])
    tol_ok = (ulp_distance(conf[live], oconf[live]) <= 1) | (np.abs(conf[live] - oconf[live]) <= 2e-2)
    assert tol_ok.all()


def test_likelihood_matches_oracle_on_hip_logits(model8b):
    """get_log_likelihood at the full widths and vocabulary (prompt 256, answer 128; 8 Monte-Carlo
    rows) against the oracle's estimator fed with the HIP forward's logits and the same masks:
    1e-3 relative (the contract); the library-RNG path is deterministic."""
    from llada_amd import get_log_likelihood
    g = np.random.Generator(np.random.PCG64(4321))
    prompt = g.integers(0, 126000, size=256, dtype=np.int64)
    answer = g.integers(0, 126000, size=128, dtype=np.int64)
    ks = [int(g.integers(1, 129)) for _ in range(2)]
    perms = [g.permutation(128) for _ in range(8)]
    k1, p1 = list(ks), iter(perms)
    val = get_log_likelihood(model8b, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=8, batch_size=4,
                             k_fn=lambda T: k1.pop(0), perm_fn=lambda T: next(p1))

    def hip_logits(ids, am):
        return model8b(torch.from_numpy(ids).cuda()).logits.float().cpu().numpy()
    k2, p2 = list(ks), iter(perms)
    ref = oll.get_log_likelihood(hip_logits, prompt, answer, mc_num=8, batch_size=4, k_fn=lambda T: k2.pop(0),
                                 perm_fn=lambda T: next(p2))
    assert abs(val - ref) <= 1e-3 * abs(ref), (val, ref)
    a = get_log_likelihood(model8b, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=64, batch_size=32, seed=3)
    b = get_log_likelihood(model8b, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=64, batch_size=32, seed=3)
    assert a == b and a < 0


def test_maximum_context_and_all_ones_mask(model8b):
    """The longest context the reference allows (4096 tokens: 3072 prompt + 1024 generated in one
    block, 128 tokens per step): exact transfer counts, and an all-ones attention_mask (the additive
    key bias is 0 everywhere) gives the same tokens as no mask at all."""
    from llada_amd import generate
    P, G, steps = 3072, 1024, 8
    prompt = prompts(1, P, seed=21)
    trace = []
    out = generate(model8b, prompt, steps=steps, gen_length=G, block_length=G, temperature=0., remasking="low_confidence",
                   step_callback=lambda b, s, x, x0, c: trace.append(int((x == MASK_ID).sum().item())))
    assert trace == [G - 128 * (i + 1) for i in range(steps)]
    assert not (out == MASK_ID).any() and torch.equal(out[0, :P].cpu(), prompt[0])
    masked = generate(model8b, prompt, torch.ones(1, P, dtype=torch.long), steps=steps, gen_length=G, block_length=G,
                      temperature=0., remasking="low_confidence")
    assert torch.equal(out, masked)


def test_empty_prompt_and_wide_batch(model8b):
    """No prompt at all (P = 0: every position starts masked) and a 64-row batch (the north-star batch on
    one GPU) with a short schedule: shapes, transfer counts, no mask token left."""
    from llada_amd import generate
    out = generate(model8b, torch.zeros((2, 0), dtype=torch.long), steps=4, gen_length=32, block_length=16, temperature=0.)
    assert tuple(out.shape) == (2, 32) and not (out == MASK_ID).any()
    prompt = prompts(64, 16, seed=5)
    counts = []
    out = generate(model8b, prompt, steps=4, gen_length=32, block_length=16, temperature=0., remasking="random", seed=3,
                   step_callback=lambda b, s, x, x0, c: counts.append((x == MASK_ID).sum(dim=1).cpu().numpy().copy()))
    assert tuple(out.shape) == (64, 48) and not (out == MASK_ID).any()
    assert np.array_equal(np.stack(counts), np.repeat(np.array([24, 16, 8, 0])[:, None], 64, axis=1))


def test_likelihood_iteration_count_edge_cases(model8b):
    """mc_num // batch_size iterations (the remainder is dropped, get_log_likelihood.py:66); none at all
    divides by zero at :77, as in the reference."""
    from llada_amd import get_log_likelihood
    g = np.random.Generator(np.random.PCG64(9))
    prompt = torch.from_numpy(g.integers(0, 126000, size=24, dtype=np.int64))
    answer = torch.from_numpy(g.integers(0, 126000, size=8, dtype=np.int64))
    a = get_log_likelihood(model8b, prompt, answer, mc_num=8, batch_size=4, seed=1)
    b = get_log_likelihood(model8b, prompt, answer, mc_num=11, batch_size=4, seed=1)     # 11 // 4 == 2 iterations too
    assert a == b
    with pytest.raises(ZeroDivisionError):
        get_log_likelihood(model8b, prompt, answer, mc_num=3, batch_size=4)
    one = get_log_likelihood(model8b, prompt, answer[:1], mc_num=4, batch_size=4, seed=2)   # single-token answer
    assert np.isfinite(one) and one < 0
