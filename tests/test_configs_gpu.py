"""The benchmark configurations at their stated sizes (LLaDA-8B shapes, random weights, a few steps each), and the
data-parallel split of both drop-ins run rank by rank on ONE GPU without a process group:

  * batched generate(), 8 and 64 prompts, gen_length 512, block_length 64 (generate.py:43-45 as
    opencompass/.../dllm.py:382-394 calls it);
  * generate() with cfg_scale 1.0 + temperature 0.7 at gen_length 1024, block_length 128 (L = 1152);
  * get_log_likelihood(mc_num=128, batch_size=32);
  * rank invariance: the rows of rank r of N see the same random streams and masks as the same rows of the
    single-rank call (generate.py:130-134 per-row independence; get_log_likelihood.py:61-77 sum over rows).
"""
import numpy as np
import pytest

from oracle import sampler as osamp
from oracle.bf16 import ulp_distance

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MASK_ID = 126336


class StopGenerate(Exception):
    pass


def prompts(B, P, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 126000, (B, P), generator=g, dtype=torch.long)


def block_logits(model, x, am, P, block_start, bl):
    """HIP forward logits of the positions [block_start, block_start + bl) of every row: [B, bl, V] float32."""
    B, L = x.shape
    rows = (torch.arange(B, dtype=torch.int32)[:, None] * L
            + torch.arange(block_start, block_start + bl, dtype=torch.int32)[None, :]).reshape(-1).cuda().contiguous()
    lg = model.forward_rows(x.cuda(), am, rows)
    return lg.float().cpu().numpy().reshape(B, bl, -1)


@pytest.mark.parametrize("B", [8, 64])
def test_batched_generate_gen512_block64(model8b, B):
    """64 synthetic prompts (and the 8-row share of one GPU of eight), gen_length 512, block_length 64: the first
    two blocks at three steps per block.  Every row reveals get_num_transfer_tokens' count (22, 21, 21) inside the
    current block; the first step's candidates and confidences equal the oracle's epilogue on the HIP forward's
    own logits (checked on two rows of the batch); the stated one-token-per-step schedule (steps = 512) starts
    with exactly one token per row."""
    from llada_amd import generate
    P, G, bl = 128, 512, 64
    prompt = prompts(B, P, seed=3)
    trace = []

    def cb(block, step, x, x0, conf):
        trace.append((block, (x == MASK_ID).sum(dim=1).cpu().numpy().copy(),
                      x0.clone() if not trace else None, conf.clone() if not trace else None, x.clone()))
        if len(trace) == 6:
            raise StopGenerate

    with pytest.raises(StopGenerate):
        generate(model8b, prompt, steps=24, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence",
                 step_callback=cb)
    left = np.stack([t[1] for t in trace])                                        # [6, B]
    want = G - np.cumsum([22, 21, 21, 22, 21, 21])
    assert np.array_equal(left, np.repeat(want[:, None], B, axis=1))
    assert [t[0] for t in trace] == [0, 0, 0, 1, 1, 1]
    for n, t in enumerate(trace):                                                 # only the current block changes
        x = t[4].cpu().numpy()
        blk = t[0]
        assert (x[:, P + (blk + 1) * bl:] == MASK_ID).all() and np.array_equal(x[:, :P], prompt.numpy())
    # oracle epilogue on the HIP logits of step 0, rows 1 and B-1 of the batch
    x_in = torch.full((B, P + G), MASK_ID, dtype=torch.long)
    x_in[:, :P] = prompt
    lg = block_logits(model8b, x_in, None, P, P, bl)
    x0, conf = trace[0][2].cpu().numpy(), trace[0][3].cpu().numpy()
    for b in (1, B - 1):
        xb = np.full((1, bl), MASK_ID, dtype=np.int64)
        ox0, oconf = osamp.confidence_step(xb, lg[b:b + 1], MASK_ID, bl, "low_confidence")
        assert np.array_equal(x0[b, P:P + bl], ox0[0])
        assert (ulp_distance(conf[b, P:P + bl], oconf[0]) <= 1).all()
        assert np.isneginf(conf[b, P + bl:]).all() and np.isneginf(conf[b, :P]).all()
    # the configuration as stated (steps == gen_length): one token per row and step
    left = []

    def cb1(block, step, x, x0, conf):
        left.append((x == MASK_ID).sum(dim=1).cpu().numpy().copy())
        if len(left) == 2:
            raise StopGenerate

    with pytest.raises(StopGenerate):
        generate(model8b, prompt, steps=G, gen_length=G, block_length=bl, temperature=0., remasking="low_confidence",
                 step_callback=cb1)
    assert np.array_equal(np.stack(left), np.repeat(np.array([G - 1, G - 2])[:, None], B, axis=1))


def test_guided_sampled_generate_gen1024_block128(model8b):
    """cfg_scale 1.0 (the forward runs the conditional and the unconditional copy: 2 x 1152 tokens) with
    temperature 0.7, gen_length 1024, block_length 128.  One step with injected float64 noise against the oracle's
    epilogue (CFG combine, float64 Gumbel-max, bf16 confidence) on the HIP forward's own logits; the library's
    Philox noise is reproducible for a seed and differs between seeds."""
    from llada_amd import generate
    P, G, bl, V = 128, 1024, 128, 126464
    L = P + G
    prompt = prompts(1, P, seed=17)
    am = torch.ones(1, P, dtype=torch.long)                       # CFG needs a mask in the reference (generate.py:86-88)
    g = np.random.Generator(np.random.PCG64(5))
    u_block = g.random(size=(1, bl, V))                           # float64 uniforms of the block's positions
    got = []

    def noise_fn(shape):
        assert tuple(shape) == (1, L, V)
        full = torch.full(shape, 0.5, dtype=torch.float64)
        full[:, P:P + bl] = torch.from_numpy(u_block)
        return full

    def cb(block, step, x, x0, conf):
        got.append((x0.cpu().numpy().copy(), conf.cpu().numpy().copy(), x.cpu().numpy().copy()))
        raise StopGenerate

    with pytest.raises(StopGenerate):
        generate(model8b, prompt, am, steps=G, gen_length=G, block_length=bl, temperature=0.7, cfg_scale=1.0,
                 remasking="low_confidence", noise_fn=noise_fn, step_callback=cb)
    x_in = torch.full((1, L), MASK_ID, dtype=torch.long)
    x_in[:, :P] = prompt
    un_in = torch.full((1, L), MASK_ID, dtype=torch.long)         # generate.py:83-84: the prompt is masked too
    am2 = torch.ones(2, L, dtype=torch.long).cuda()
    lg = block_logits(model8b, torch.cat([x_in, un_in]), am2, P, P, bl)          # [2, bl, V]
    logits = osamp.cfg_combine(lg[:1], lg[1:], 1.0)
    xb = np.full((1, bl), MASK_ID, dtype=np.int64)
    ox0, oconf = osamp.confidence_step(xb, logits, MASK_ID, bl, "low_confidence", temperature=0.7, u=u_block)
    x0, conf, x_after = got[0]
    assert np.array_equal(x0[0, P:P + bl], ox0[0])
    assert (ulp_distance(conf[0, P:P + bl], oconf[0]) <= 1).all()
    # exactly one token was transferred, the most confident one (lowest index on a tie)
    changed = np.flatnonzero(x_after[0] != x_in[0].numpy())
    assert len(changed) == 1 and changed[0] - P == int(np.flatnonzero(oconf[0] == oconf[0].max())[0])
    assert x_after[0, changed[0]] == ox0[0, changed[0] - P]

    def run(seed, n=3):
        xs = []

        def cbn(block, step, x, x0, conf):
            xs.append(torch.stack([x, x0]))     # the committed tokens and every position's sampled candidate
            if len(xs) == n:
                raise StopGenerate
        with pytest.raises(StopGenerate):
            generate(model8b, prompt, am, steps=G, gen_length=G, block_length=bl, temperature=0.7, cfg_scale=1.0,
                     remasking="low_confidence", seed=seed, step_callback=cbn)
        return torch.stack(xs)
    a, b, c = run(11), run(11), run(12)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert ((a[:, 0] == MASK_ID).sum(dim=(1, 2)).cpu().numpy() == G - np.arange(1, 4)).all()


def test_likelihood_mc128_batch32(model8b):
    """get_log_likelihood(mc_num=128, batch_size=32) on a 256-token prompt and a 128-token answer: finite, negative,
    bit-identical across two runs; and the two-rank split of every iteration's 32 rows, run rank by rank on this
    GPU, adds up to the single-rank sums (the masks are keyed by the global row; only the GEMM tiling of 16
    instead of 32 rows differs, i.e. bf16 rounding noise)."""
    from llada_amd import get_log_likelihood
    g = np.random.Generator(np.random.PCG64(4321))
    prompt = torch.from_numpy(g.integers(0, 126000, size=256, dtype=np.int64))
    answer = torch.from_numpy(g.integers(0, 126000, size=128, dtype=np.int64))
    a = get_log_likelihood(model8b, prompt, answer, mc_num=128, batch_size=32, seed=7)
    b = get_log_likelihood(model8b, prompt, answer, mc_num=128, batch_size=32, seed=7)
    assert np.isfinite(a) and a < 0 and a == b
    whole = get_log_likelihood(model8b, prompt, answer, mc_num=128, batch_size=32, seed=7, return_loss_sums=True)
    parts = [get_log_likelihood(model8b, prompt, answer, mc_num=128, batch_size=32, seed=7, rank=r, world_size=2,
                                return_loss_sums=True) for r in range(2)]
    whole, summed = whole.cpu().numpy(), (parts[0] + parts[1]).cpu().numpy()
    assert whole.shape == (4,) and (parts[0].cpu().numpy() > 0).all() and (parts[1].cpu().numpy() > 0).all()
    rel = np.abs(summed - whole) / np.abs(whole)
    assert rel.max() <= 1e-4, rel
    assert abs(-(whole / 32).mean() - a) <= 1e-12 * abs(a)


def test_likelihood_mc128_batch32_eight_rank_shares(model8b):
    """The same call split over 8 ranks, run rank by rank on this GPU: every rank holds 4 of the 32 rows of each of
    the 4 iterations and runs its 16 rows through ONE forward (fuse_iterations).  The eight ranks' per-iteration sums
    add up to the single-rank sums within 1e-4 (GEMM tilings of 16 x 384 instead of 32 x 384 rows), and one rank's
    fused sums agree with its own per-iteration loop to the same bound."""
    from llada_amd import get_log_likelihood
    g = np.random.Generator(np.random.PCG64(4321))
    prompt = torch.from_numpy(g.integers(0, 126000, size=256, dtype=np.int64))
    answer = torch.from_numpy(g.integers(0, 126000, size=128, dtype=np.int64))
    kw = dict(mc_num=128, batch_size=32, seed=7, return_loss_sums=True)
    whole = get_log_likelihood(model8b, prompt, answer, **kw).cpu().numpy()
    parts = [get_log_likelihood(model8b, prompt, answer, rank=r, world_size=8, **kw).cpu().numpy() for r in range(8)]
    rel = np.abs(sum(parts) - whole) / np.abs(whole)
    assert rel.max() <= 1e-4, rel
    loop3 = get_log_likelihood(model8b, prompt, answer, rank=3, world_size=8, fuse_iterations=False, **kw).cpu().numpy()
    assert (np.abs(parts[3] - loop3) / np.abs(loop3)).max() <= 1e-4


class TableModel:
    """Protocol-only model (generate.py:60,88,92) whose logits depend on nothing but the token and its position, so
    a row's logits are the same whatever batch it sits in: what is left to differ between rank counts is the
    library's own per-row state (random streams, masks, schedules)."""

    def __init__(self, vocab=2048, max_len=256, seed=0):
        self.device = torch.device("cuda")
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.tok = (torch.randn(vocab, vocab, device="cuda", generator=g) * 2).to(torch.bfloat16)
        self.pos = (torch.randn(max_len, vocab, device="cuda", generator=g) * 2).to(torch.bfloat16)

    def __call__(self, x, attention_mask=None):
        class Out:
            pass
        o = Out()
        o.logits = self.tok[x] + self.pos[None, :x.shape[1]]
        return o


@pytest.mark.parametrize("remasking,temperature", [("low_confidence", 0.7), ("random", 0.0), ("random", 0.7)])
def test_generate_row_shards_match_the_whole_batch(lib, remasking, temperature):
    """generate() on rows 4..8 with row_offset=4 (and on three uneven shards) returns exactly the rows the 8-row call
    returns: Gumbel noise and 'random' scores are keyed by the global row, the schedule and the top-k are per row."""
    from llada_amd import generate
    from llada_amd.parallel import shard_bounds
    mask_id = 2047
    model = TableModel()
    prompt = torch.randint(0, 2000, (8, 24), generator=torch.Generator().manual_seed(2))
    kw = dict(steps=16, gen_length=32, block_length=16, temperature=temperature, remasking=remasking, mask_id=mask_id, seed=9)
    whole = generate(model, prompt, **kw)
    assert not (whole == mask_id).any()
    half = generate(model, prompt[4:8], row_offset=4, **kw)
    assert torch.equal(half, whole[4:8])
    bounds = shard_bounds(8, 3)
    pieces = [generate(model, prompt[bounds[r]:bounds[r + 1]], row_offset=bounds[r], **kw) for r in range(3)]
    assert torch.equal(torch.cat(pieces), whole)
    # step by step too (the table model's temperature-0 tokens do not depend on the order of unmasking, the
    # intermediate states do), and the wrong offset draws other streams
    def states(p, off):
        xs = []
        generate(model, p, row_offset=off, step_callback=lambda b, s, x, x0, c: xs.append(x.clone()), **kw)
        return torch.stack(xs)
    mid = states(prompt, 0)
    assert torch.equal(states(prompt[4:8], 4), mid[:, 4:8])
    assert not torch.equal(states(prompt[4:8], 0), mid[:, 4:8])


def test_likelihood_rank_sums_add_up_exactly(lib):
    """The per-rank loss sums of get_log_likelihood(rank=, world_size=) for 1, 2, 3 and 5 ranks on a model whose
    logits do not depend on the batch: the masks (Philox, global row) and p_mask are those of the single-rank call,
    so the sums agree to float64 rounding, with and without classifier-free guidance."""
    from llada_amd import get_log_likelihood
    mask_id = 2047
    model = TableModel(seed=1)
    g = torch.Generator().manual_seed(4)
    prompt = torch.randint(0, 2000, (40,), generator=g)
    answer = torch.randint(0, 2000, (24,), generator=g)
    for cfg in (0.0, 1.5):
        kw = dict(mc_num=32, batch_size=8, cfg_scale=cfg, mask_id=mask_id, seed=3, return_loss_sums=True)
        whole = get_log_likelihood(model, prompt, answer, **kw).cpu().numpy()
        assert whole.shape == (4,) and (whole > 0).all()
        for world in (2, 3, 5, 8):
            parts = sum(get_log_likelihood(model, prompt, answer, rank=r, world_size=world, **kw).cpu().numpy()
                        for r in range(world))
            assert np.abs(parts - whole).max() <= 1e-9 * np.abs(whole).max(), (world, parts, whole)
