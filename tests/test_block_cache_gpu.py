"""The block-cached ("dual cache") sampler step: an opt-in approximation with no counterpart in the
reference (generate.py:80-92 recomputes every position at every step; README.md:110-116, 127-130 point to
caching as the missing speed-up).  Checked against the oracle's restatement of the same approximation
(oracle/model.py::forward_block) and through exactness properties that tie it to the uncached path."""
import numpy as np
import pytest

from oracle import sampler as osamp
from oracle.bf16 import round_bf16, ulp_distance
from oracle.model import LLaDAConfig as OCfg
from oracle.model import forward, forward_block, init_weights

MASK_ID = 511

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def build(ocfg, W, packed=True):
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    cfg = LLaDAConfig(d_model=ocfg.d_model, n_layers=ocfg.n_layers, n_heads=ocfg.n_heads, mlp_hidden=ocfg.mlp_hidden,
                      vocab_size=ocfg.vocab_size, rope_theta=ocfg.rope_theta, rms_eps=ocfg.rms_eps)
    return LLaDAModel(LLaDAWeights.from_numpy(cfg, W, "cuda", packed=packed))


def max_rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def small():
    ocfg = OCfg(d_model=256, n_layers=3, n_heads=2, mlp_hidden=512, vocab_size=512)
    return ocfg, init_weights(ocfg, seed=21, std=0.05, head_scale=4.0)


@pytest.mark.parametrize("n_seq,L,H,q0,n_q,masked", [(2, 384, 2, 128, 32, False), (3, 200, 2, 150, 50, True),
                                                     (1, 513, 1, 0, 129, True), (2, 100, 1, 99, 1, False)])
def test_attention_query_subset_is_exact(lib, n_seq, L, H, q0, n_q, masked):
    """Queries q0 .. q0+n_q-1 of every sequence against all keys: the same numbers, bit for bit, as
    those rows of the all-queries kernel (a query's result does not depend on its neighbours)."""
    d = H * 128
    g = np.random.Generator(np.random.PCG64(L + n_q))
    qkv = torch.from_numpy(round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))).cuda().to(torch.bfloat16)
    km = None
    if masked:
        kmn = np.ones((n_seq, L), dtype=np.uint8)
        for b in range(n_seq):
            kmn[b, : 5 + 11 * b] = 0
        km = torch.from_numpy(kmn).cuda()
    st = torch.cuda.current_stream().cuda_stream
    full = torch.empty((n_seq * L, d), dtype=torch.bfloat16, device="cuda")
    assert lib.llada_attn_bidir_bf16(qkv.data_ptr(), None if km is None else km.data_ptr(), full.data_ptr(), n_seq, L, H,
                                     st) == 0, lib.llada_last_error()
    # compact query rows with a stride wider than the heads (as the block's [Q | K | V] rows have)
    q = qkv.reshape(n_seq, L, 3 * d)[:, q0:q0 + n_q].contiguous().reshape(n_seq * n_q, 3 * d)
    sub = torch.full((n_seq * n_q, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    assert lib.llada_attn_bidir_queries_bf16(q.data_ptr(), 3 * d, n_q, qkv.data_ptr(), None if km is None else km.data_ptr(),
                                             sub.data_ptr(), n_seq, L, H, None, 0, None, 0, st) == 0, lib.llada_last_error()
    torch.cuda.synchronize()
    want = full.reshape(n_seq, L, d)[:, q0:q0 + n_q].reshape(n_seq * n_q, d)
    assert torch.equal(sub, want)


def test_attention_queries_rejects_bad_arguments(lib):
    t = torch.zeros((64, 384), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.llada_attn_bidir_queries_bf16(t.data_ptr(), 64, 8, t.data_ptr(), None, t.data_ptr(), 1, 64, 1, None, 0, None, 0, st) == -1
    assert lib.llada_attn_bidir_queries_bf16(t.data_ptr(), 384, 0, t.data_ptr(), None, t.data_ptr(), 1, 64, 1, None, 0, None, 0, st) == -1
    assert lib.llada_attn_bidir_queries_bf16(None, 384, 8, t.data_ptr(), None, t.data_ptr(), 1, 64, 1, None, 0, None, 0, st) == -1


@pytest.mark.parametrize("splits", [0, 2, 3, 16])
@pytest.mark.parametrize("n_seq,L,H,q0,n_q,masked", [(1, 384, 2, 128, 32, False), (2, 384, 2, 352, 32, True),
                                                     (2, 1152, 1, 128, 128, True), (1, 200, 2, 60, 70, True),
                                                     (8, 384, 4, 160, 32, False)])
def test_attention_key_split(lib, n_seq, L, H, q0, n_q, masked, splits):
    """Few queries over many keys (the block-cached step): the keys are cut into runs of tiles, one workgroup
    each, and a second kernel merges the partial softmaxes.  Same result as the one-pass kernel up to the
    float32 summation order (checked to one bf16 step of the largest output) and within the attention
    tolerance of the oracle; includes a run of keys that is masked out entirely."""
    from oracle.model import attention as oattn
    d = H * 128
    g = np.random.Generator(np.random.PCG64(L + n_q + splits))
    qkv_n = round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))
    qkv_n[L // 2, d:2 * d] *= 4
    qkv = torch.from_numpy(qkv_n).cuda().to(torch.bfloat16)
    km = kmn = None
    if masked:
        kmn = np.ones((n_seq, L), dtype=np.uint8)
        kmn[:, :70] = 0                    # the whole first 64-key tile (a split with nothing to attend)
        kmn[n_seq - 1, 100:131] = 0
        km = torch.from_numpy(kmn).cuda()
    st = torch.cuda.current_stream().cuda_stream
    q = qkv.reshape(n_seq, L, 3 * d)[:, q0:q0 + n_q].contiguous().reshape(n_seq * n_q, 3 * d)
    kmp = None if km is None else km.data_ptr()
    one = torch.full((n_seq * n_q, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    assert lib.llada_attn_bidir_queries_bf16(q.data_ptr(), 3 * d, n_q, qkv.data_ptr(), kmp, one.data_ptr(), n_seq, L, H,
                                             None, 0, None, 0, st) == 0, lib.llada_last_error()
    need = lib.llada_attn_workspace_bytes(n_seq, n_q, L, H)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    two = torch.full((n_seq * n_q, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    lib.llada_attn_force_key_splits(splits)
    try:
        assert lib.llada_attn_bidir_queries_bf16(q.data_ptr(), 3 * d, n_q, qkv.data_ptr(), kmp, two.data_ptr(), n_seq, L,
                                                 H, None, 0, ws.data_ptr(), ws.numel(), st) == 0, lib.llada_last_error()
        torch.cuda.synchronize()
    finally:
        lib.llada_attn_force_key_splits(0)
    a, b = one.float().cpu().numpy(), two.float().cpu().numpy()
    assert np.isfinite(b).all()
    assert np.abs(a - b).max() <= 2.0 ** -7 * np.abs(a).max()
    x = qkv_n.reshape(n_seq, L, 3, H, 128)
    qq, kk, vv = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    ref = oattn(qq[:, :, q0:q0 + n_q], kk, vv, kmn, False).transpose(0, 2, 1, 3).reshape(n_seq * n_q, d)
    assert np.abs(b - ref).max() <= 1.5e-2 * np.abs(ref).max()


@pytest.mark.parametrize("packed", [True, False])
def test_fill_cache_leaves_logits_unchanged(lib, small, packed):
    ocfg, W = small
    model = build(ocfg, W, packed)
    g = np.random.Generator(np.random.PCG64(5))
    ids = torch.from_numpy(g.integers(0, 512, size=(2, 40), dtype=np.int64)).cuda()
    rows = torch.arange(80, dtype=torch.int32, device="cuda")
    plain = model.forward_rows(ids, None, rows)
    kv = model.kv_cache(2, 40)
    cached = model.forward_rows(ids, None, rows, kv_cache=kv)
    assert torch.equal(plain, cached)


@pytest.mark.parametrize("packed", [True, False])
def test_block_step_on_unchanged_ids_reproduces_the_full_forward(lib, small, packed):
    """Right after the cache was filled, with the block's tokens unchanged, the block step sees exactly the
    keys and values the full forward saw, so its logits are the full forward's rows for the block (up to the
    accumulation order of the smaller GEMMs)."""
    ocfg, W = small
    model = build(ocfg, W, packed)
    g = np.random.Generator(np.random.PCG64(6))
    n_seq, L, bs, bl = 3, 48, 16, 16
    idn = g.integers(0, 512, size=(n_seq, L), dtype=np.int64)
    km = np.ones((n_seq, L), dtype=np.int64)
    km[1, :5] = 0
    ids, am = torch.from_numpy(idn).cuda(), torch.from_numpy(km).cuda()
    kv = model.kv_cache(n_seq, L)
    full = model.forward_rows(ids, am, None, kv_cache=kv).reshape(n_seq, L, -1)[:, bs:bs + bl].reshape(n_seq * bl, -1)
    blk = model.forward_block(ids[:, bs:bs + bl], am, L, bs, kv)
    a, b = blk.float().cpu().numpy(), full.float().cpu().numpy()
    assert max_rel(a, b) <= 2.0 ** -7, max_rel(a, b)
    assert (ulp_distance(a, b) <= 1).mean() >= 0.95


def test_block_step_matches_oracle(lib, small):
    """Changed block tokens, two steps in a row (the second sees the keys/values the first wrote), key
    padding: logits within the forward's bf16 tolerance of the oracle's restatement."""
    ocfg, W = small
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(7))
    n_seq, L, bs, bl = 2, 40, 24, 8
    idn = g.integers(0, 511, size=(n_seq, L), dtype=np.int64)
    idn[:, bs:] = MASK_ID
    km = np.ones((n_seq, L), dtype=np.int64)
    km[0, :3] = 0
    am = torch.from_numpy(km).cuda()
    kv = model.kv_cache(n_seq, L)
    okv = []
    got = model.forward_rows(torch.from_numpy(idn).cuda(), am, None, kv_cache=kv).float().cpu().numpy()
    ref = forward(W, ocfg, idn, km, emulate_bf16=True, kv_out=okv).reshape(n_seq * L, -1)
    assert max_rel(got, ref) <= 2e-2
    blk = idn[:, bs:bs + bl].copy()
    for step in range(2):
        blk[:, step * 3: step * 3 + 3] = g.integers(0, 511, size=(n_seq, 3))
        got = model.forward_block(torch.from_numpy(blk).cuda(), am, L, bs, kv).float().cpu().numpy()
        ref = forward_block(W, ocfg, blk, bs, okv, km, emulate_bf16=True).reshape(n_seq * bl, -1)
        assert max_rel(got, ref) <= 2e-2, (step, max_rel(got, ref))
    # and the cached forward really differs from a fresh full forward (it is an approximation: the rest of the
    # sequence does not see the new block tokens), so the test above is not vacuous
    idn2 = idn.copy()
    idn2[:, bs:bs + bl] = blk
    fresh = forward(W, ocfg, idn2, km, emulate_bf16=True)[:, bs:bs + bl].reshape(n_seq * bl, -1)
    assert max_rel(ref, fresh) > 1e-3


def test_block_step_without_cache_update_gives_the_same_logits(lib, small):
    """update_cache=False reads the block's fresh keys/values from the compact rows instead of scattering them
    into the image first: the attention sees the same bits, so the logits are identical; the cache is untouched."""
    ocfg, W = small
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(17))
    n_seq, L, bs, bl = 3, 200, 100, 36           # the block straddles key tiles (64): 100..135
    ids = torch.from_numpy(g.integers(0, 511, size=(n_seq, L), dtype=np.int64)).cuda()
    km = np.ones((n_seq, L), dtype=np.int64)
    km[2, :7] = 0
    am = torch.from_numpy(km).cuda()
    kv_a, kv_b = model.kv_cache(n_seq, L), model.kv_cache(n_seq, L)
    model.forward_rows(ids, am, None, kv_cache=kv_a)
    kv_b.copy_(kv_a)
    blk = torch.from_numpy(g.integers(0, 511, size=(n_seq, bl), dtype=np.int64)).cuda()
    a = model.forward_block(blk, am, L, bs, kv_a, update_cache=True).clone()
    b = model.forward_block(blk, am, L, bs, kv_b, update_cache=False).clone()
    assert torch.equal(a, b)
    assert not torch.equal(kv_a, kv_b)            # only the first call wrote the block's keys/values
    kv_ref = model.kv_cache(n_seq, L)
    model.forward_rows(ids, am, None, kv_cache=kv_ref)
    assert torch.equal(kv_b, kv_ref)


def test_block_step_rejects_bad_arguments(lib, small):
    ocfg, W = small
    model = build(ocfg, W)
    ids = torch.zeros((1, 8), dtype=torch.long, device="cuda")
    kv = model.kv_cache(1, 16)
    with pytest.raises(AssertionError):
        model.forward_block(ids, None, 16, 12, kv)            # block runs past the sequence
    with pytest.raises(AssertionError):
        model.forward_block(ids, None, 32, 0, kv)             # cache too small for the sequence


@pytest.mark.parametrize("cache", ["dual", "prefix"])
def test_generate_dual_cache_single_step_blocks_equal_uncached(lib, small, cache):
    """One step per block: only full forwards run, so the cached sampler is the uncached one, bit for bit."""
    from llada_amd import generate
    ocfg, W = small
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(8))
    prompt = torch.from_numpy(g.integers(0, MASK_ID, size=(2, 9), dtype=np.int64))
    a = generate(model, prompt, steps=4, gen_length=32, block_length=8, mask_id=MASK_ID)
    b = generate(model, prompt, steps=4, gen_length=32, block_length=8, mask_id=MASK_ID, cache=cache)
    assert torch.equal(a, b)


def test_prefix_cache_step_matches_oracle_and_row_subset_is_exact(lib, small):
    """The "prefix" flavour of the cached step: everything from the block's start to the end of the sequence is
    recomputed against the cached keys / values of the positions left of it (llada_forward_block with a range that
    reaches the end, logits for the block's rows only).  Against the oracle's restatement (forward_block over the
    same range) to the bf16 bound; the row subset gives exactly the rows of the all-rows call."""
    ocfg, W = small
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(12))
    B, L, bs, bl = 2, 40, 16, 8
    ids0 = g.integers(0, MASK_ID, size=(B, L), dtype=np.int64)
    ids0[:, bs:] = MASK_ID
    am = np.ones((B, L), dtype=np.int64)
    am[1, :3] = 0
    kv = model.kv_cache(B, L)
    model.forward_rows(torch.from_numpy(ids0).cuda(), torch.from_numpy(am).cuda(), None, kv_cache=kv)
    okv = []
    forward(W, ocfg, ids0, am, emulate_bf16=True, kv_out=okv)
    ids1 = ids0.copy()
    ids1[:, bs:bs + 3] = g.integers(0, MASK_ID, size=(B, 3))            # three positions of the block got revealed
    tail = torch.from_numpy(ids1[:, bs:]).cuda()
    rows = (torch.arange(B, dtype=torch.int32)[:, None] * (L - bs) + torch.arange(bl, dtype=torch.int32)[None, :]).reshape(-1).cuda()
    mask_t = torch.from_numpy(am).cuda()
    sub = model.forward_block(tail, mask_t, L, bs, kv, update_cache=False, logit_rows=rows.contiguous())
    full = model.forward_block(tail, mask_t, L, bs, kv, update_cache=False)
    assert tuple(sub.shape) == (B * bl, ocfg.vocab_size)
    assert torch.equal(sub, full.reshape(B, L - bs, -1)[:, :bl].reshape(B * bl, -1))
    ref = forward_block(W, ocfg, ids1[:, bs:], bs, okv, am, emulate_bf16=True)[:, :bl]
    assert max_rel(sub.float().cpu().numpy().reshape(B, bl, -1), ref) <= 2e-2


@pytest.mark.parametrize("cache", ["dual", "prefix"])
@pytest.mark.parametrize("cfg_scale,remasking", [(0.0, "low_confidence"), (1.0, "low_confidence"), (0.0, "random")])
def test_generate_dual_cache_teacher_forced(lib, small, cfg_scale, remasking, cache):
    """Every step of generate(cache='dual'): the oracle's epilogue on the logits the oracle's cached forward
    gives for that step's input must name the same transfers (given logits that agree to bf16 tolerance, checked
    through the confidences), exact transfer counts per step, monotone unmasking inside the current block."""
    from llada_amd import generate
    ocfg, W = small
    model = build(ocfg, W)
    g = np.random.Generator(np.random.PCG64(9))
    B, P, G, bl, steps = 2, 8, 16, 8, 8
    prompt = g.integers(0, MASK_ID, size=(B, P), dtype=np.int64)
    am = np.ones((B, P), dtype=np.int64)
    am[1, :2] = 0
    trace = []

    def cb(block, step, x, x0, conf):
        trace.append((block, step, x.cpu().numpy().copy(), x0.cpu().numpy().copy(), conf.cpu().numpy().copy()))

    out = generate(model, torch.from_numpy(prompt), torch.from_numpy(am), steps=steps, gen_length=G, block_length=bl,
                   cfg_scale=cfg_scale, remasking=remasking, mask_id=MASK_ID, step_callback=cb, cache=cache, seed=3)
    out = out.cpu().numpy()
    assert np.array_equal(out[:, :P], prompt) and not (out == MASK_ID).any()
    x_in = np.full((B, P + G), MASK_ID, dtype=np.int64)
    x_in[:, :P] = prompt
    spb = steps // (G // bl)
    L, V = P + G, ocfg.vocab_size
    full_mask = torch.from_numpy(np.concatenate([am, np.ones((B, G), np.int64)], 1)).cuda()
    kv2 = model.kv_cache(B, L)
    for block, step, x_after, x0, conf in trace:
        bs, be = P + block * bl, P + (block + 1) * bl
        changed = x_after != x_in
        if cfg_scale == 0.0 and remasking == "low_confidence":
            # the same forward calls generate() makes, on a second cache: identical logits (the kernels are
            # deterministic), then the oracle's epilogue on them
            xt = torch.from_numpy(x_in).cuda()
            if step == 0:
                rows = (torch.arange(B, dtype=torch.int32, device="cuda")[:, None] * L
                        + torch.arange(bs, be, dtype=torch.int32, device="cuda")[None, :]).reshape(-1).contiguous()
                lg = model.forward_rows(xt, full_mask, rows, kv_cache=kv2).reshape(B, bl, V)
            elif cache == "dual":
                lg = model.forward_block(xt[:, bs:be], full_mask, L, bs, kv2).reshape(B, bl, V)
            else:
                crows = (torch.arange(B, dtype=torch.int32, device="cuda")[:, None] * (L - bs)
                         + torch.arange(bl, dtype=torch.int32, device="cuda")[None, :]).reshape(-1).contiguous()
                lg = model.forward_block(xt[:, bs:], full_mask, L, bs, kv2, update_cache=False, logit_rows=crows).reshape(B, bl, V)
            lg_full = np.zeros((B, L, V), dtype=np.float32)
            lg_full[:, bs:be] = lg.float().cpu().numpy()
            ox0, oconf = osamp.confidence_step(x_in, lg_full, MASK_ID, be, "low_confidence")
            live = ~np.isneginf(oconf)
            assert np.array_equal(np.isneginf(conf), np.isneginf(oconf))
            assert np.array_equal(x0[live], ox0[live])
            assert ulp_distance(conf[live], oconf[live]).max(initial=0) <= 1
        k = osamp.get_num_transfer_tokens(x_in[:, bs:be] == MASK_ID, spb - step)[:, 0]
        assert np.array_equal(changed.sum(1), k)                       # exact transfer counts
        assert not changed[:, :bs].any() and not changed[:, be:].any()  # only inside the current block
        assert (x_in[changed] == MASK_ID).all()                         # monotone: only masked positions change
        assert np.array_equal(x_after[changed], x0[changed])
        for j in range(B):
            assert osamp.is_valid_topk(conf[j], changed[j], changed[j].sum())
        x_in = x_after
    assert np.array_equal(x_in, out)
    # determinism
    again = generate(model, torch.from_numpy(prompt), torch.from_numpy(am), steps=steps, gen_length=G, block_length=bl,
                     cfg_scale=cfg_scale, remasking=remasking, mask_id=MASK_ID, cache=cache, seed=3).cpu().numpy()
    assert np.array_equal(out, again)


def test_generate_dual_cache_needs_the_hip_model(lib):
    from llada_amd import generate

    class Foreign:
        device = "cuda"

    with pytest.raises(NotImplementedError):
        generate(Foreign(), torch.zeros((1, 4), dtype=torch.long), steps=4, gen_length=4, block_length=4, cache="dual")
    with pytest.raises(NotImplementedError):
        generate(Foreign(), torch.zeros((1, 4), dtype=torch.long), steps=4, gen_length=4, block_length=4, cache="prefix")
