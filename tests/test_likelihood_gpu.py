"""Parity of the likelihood path (mask generation, masked cross-entropy, get_log_likelihood)
against the oracle and the golden vectors captured from the reference, on a real MI355X."""
import numpy as np
import pytest

from oracle import likelihood as oll
from oracle import rng as orng
from tests.golden_util import FakeModel, MASK_ID, fake_model_logits, load

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(lib, rc):
    assert rc == 0, lib.llada_last_error().decode()


def _ll_cases():
    return load("likelihood.npz")[1]


class TorchFakeModel:
    def __init__(self, seed, vocab):
        self.device = torch.device("cuda")
        self.seed, self.vocab, self.calls = seed, vocab, 0

    def __call__(self, x, attention_mask=None):
        lg = fake_model_logits(self.seed, self.calls, tuple(x.shape) + (self.vocab,))
        self.calls += 1

        class Out:
            pass
        o = Out()
        o.logits = dev(lg, torch.bfloat16)
        return o


@pytest.mark.parametrize("case", _ll_cases(), ids=lambda c: c["name"])
def test_ll_mask_matches_reference(lib, case):
    """forward_process with the reference's recorded k and permutations gives the reference's
    perturbed sequences bit for bit."""
    from llada_amd.get_log_likelihood import stratified_counts
    z, _ = load("likelihood.npz")
    n = case["name"]
    b, l1, l2 = case["b"], case["l1"], case["l2"]
    seq = np.repeat(np.concatenate([z[f"{n}/prompt"], z[f"{n}/answer"]])[None], b, axis=0)
    for it, k in enumerate(z[f"{n}/k"]):
        counts = stratified_counts(int(k), b, l2).astype(np.int32)
        noisy = torch.empty((b, l1 + l2), dtype=torch.long, device="cuda")
        st, ct, pt = dev(seq), dev(counts), dev(z[f"{n}/perms"][it].astype(np.int32))
        check(lib, lib.llada_ll_mask(st.data_ptr(), noisy.data_ptr(), b, l1 + l2, l1, MASK_ID, ct.data_ptr(),
                                     pt.data_ptr(), 0, it, 0, stream()))
        torch.cuda.synchronize()
        assert np.array_equal(noisy.cpu().numpy(), z[f"{n}/perturbed"][it])


@pytest.mark.parametrize("case", _ll_cases(), ids=lambda c: c["name"])
def test_get_log_likelihood_matches_reference(lib, case):
    """The drop-in with injected k/permutations and a foreign model reproduces the reference's
    estimate (contract: 1e-3 relative; observed far tighter)."""
    from llada_amd import get_log_likelihood
    z, _ = load("likelihood.npz")
    n = case["name"]
    ks = list(z[f"{n}/k"])
    it = iter(z[f"{n}/perms"].reshape(-1, case["l2"]))
    val = get_log_likelihood(TorchFakeModel(case["seed"], case["V"]), torch.from_numpy(z[f"{n}/prompt"]),
                             torch.from_numpy(z[f"{n}/answer"]), mc_num=case["mc"], batch_size=case["b"],
                             cfg_scale=case["cfg"], mask_id=MASK_ID, k_fn=lambda T: int(ks.pop(0)),
                             perm_fn=lambda T: next(it))
    assert isinstance(val, float)
    ref = float(z[f"{n}/value"])
    assert abs(val - ref) <= 1e-3 * abs(ref), (val, ref)
    # against the oracle (GPU-style float32 log-sum-exp, no torch-CPU quirk): essentially exact
    ks2 = list(z[f"{n}/k"])
    it2 = iter(z[f"{n}/perms"].reshape(-1, case["l2"]))
    oval = oll.get_log_likelihood(FakeModel(case["seed"], case["V"]), z[f"{n}/prompt"], z[f"{n}/answer"],
                                  mc_num=case["mc"], batch_size=case["b"], cfg_scale=case["cfg"], mask_id=MASK_ID,
                                  k_fn=lambda T: int(ks2.pop(0)), perm_fn=lambda T: next(it2).astype(np.int64))
    assert abs(val - oval) <= 2e-4 * abs(oval), (val, oval)


@pytest.mark.parametrize("splits", [None, 1, 3], ids=["whole_row", "stream_1", "stream_3"])
def test_masked_ce_float32_mode(lib, splits):
    """round_ce_bf16 = 0 keeps float32 cross-entropies; compare with a float64 evaluation (the whole-row
    kernel, and the streamed row pieces with one and three pieces per row)."""
    b, l, l1, V = 3, 20, 6, 512
    g = np.random.Generator(np.random.PCG64(3))
    seq = g.integers(0, V, size=(b, l), dtype=np.int64)
    noisy = seq.copy()
    noisy[:, l1:][g.random((b, l - l1)) < 0.5] = MASK_ID
    lg = fake_model_logits(5, 0, (b, l, V))
    pm = np.array([0.25, 0.5, 1.0], dtype=np.float32)
    rows = (np.arange(b)[:, None] * l + np.arange(l1, l)[None, :]).reshape(-1).astype(np.int32)
    row_loss = torch.empty(rows.size, dtype=torch.float32, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    lgt = dev(lg.reshape(b * l, V)[rows], torch.bfloat16)
    rt, nt, st, pt = dev(rows), dev(noisy), dev(seq), dev(pm)
    ws = None if splits is None else torch.empty(lib.llada_rows_workspace_bytes(rows.size), dtype=torch.uint8, device="cuda")
    lib.llada_rows_force_splits(splits or 0)
    try:
        check(lib, lib.llada_masked_ce(lgt.data_ptr(), None, 0.0, rt.data_ptr(), rows.size, V, nt.data_ptr(), st.data_ptr(),
                                       b, l, MASK_ID, pt.data_ptr(), 0, row_loss.data_ptr(), out.data_ptr(),
                                       None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), stream()))
        torch.cuda.synchronize()
    finally:
        lib.llada_rows_force_splits(0)
    lp = lg.astype(np.float64)
    lp = lp - np.log(np.exp(lp - lp.max(-1, keepdims=True)).sum(-1, keepdims=True)) - lp.max(-1, keepdims=True)
    ce = -np.take_along_axis(lp, seq[..., None], axis=-1)[..., 0]
    want = (np.where(noisy == MASK_ID, ce, 0.0) / pm[:, None].astype(np.float64)).sum()
    assert abs(out.item() - want) <= 1e-5 * abs(want)


def test_ll_mask_philox(lib):
    """Without injected permutations the device masks the counts[r] smallest Philox keys of each row."""
    b, l, l1 = 5, 60, 13
    T = l - l1
    seed, it, row_offset = 99, 2, 4
    g = np.random.Generator(np.random.PCG64(4))
    seq = g.integers(0, 1000, size=(b, l), dtype=np.int64)
    counts = np.array([1, 7, T, 20, 33], dtype=np.int32)
    noisy = torch.empty((b, l), dtype=torch.long, device="cuda")
    st, ct = dev(seq), dev(counts)
    check(lib, lib.llada_ll_mask(st.data_ptr(), noisy.data_ptr(), b, l, l1, MASK_ID, ct.data_ptr(), None, seed, it,
                                 row_offset, stream()))
    torch.cuda.synchronize()
    keys = orng.ll_sort_keys(seed, it, row_offset + np.arange(b), T)
    want = seq.copy()
    for r in range(b):
        order = np.lexsort((np.arange(T), keys[r]))
        want[r, l1 + order[: counts[r]]] = MASK_ID
    assert np.array_equal(noisy.cpu().numpy(), want)
    assert ((noisy.cpu().numpy() == MASK_ID).sum(axis=1) == counts).all()


def test_ll_masked_rows(lib):
    """Device compaction of the masked answer positions == numpy's row-major nonzero, for ragged counts
    (including a row masked completely and answers longer than one 256-thread trip)."""
    b, l, l1 = 6, 700, 37
    T = l - l1
    g = np.random.Generator(np.random.PCG64(8))
    seq = g.integers(0, 1000, size=(b, l), dtype=np.int64)
    counts = np.array([1, T, 300, 17, 256, 511], dtype=np.int32)
    noisy = seq.copy()
    for r in range(b):
        noisy[r, l1 + g.permutation(T)[: counts[r]]] = MASK_ID
    out = torch.full((int(counts.sum()) + 8,), -1, dtype=torch.int32, device="cuda")
    nt, ct = dev(noisy), dev(counts)
    check(lib, lib.llada_ll_masked_rows(nt.data_ptr(), b, l, l1, MASK_ID, ct.data_ptr(), out.data_ptr(), stream()))
    torch.cuda.synchronize()
    want = np.flatnonzero(noisy.reshape(-1) == MASK_ID).astype(np.int32)
    got = out.cpu().numpy()
    assert np.array_equal(got[: want.size], want)
    assert (got[want.size:] == -1).all()


def test_fused_iterations_give_the_per_iteration_sums(lib):
    """A rank that holds few rows runs all its Monte-Carlo iterations through one forward
    (get_log_likelihood(fuse_iterations=True), the default): masks, masked positions and p_mask are those of the
    per-iteration loop (get_log_likelihood.py:66-75), so each iteration's loss sum agrees with the loop's up to the
    bf16 rounding noise of another GEMM tiling -- with and without guidance, and as one rank of a split batch."""
    import torch
    from llada_amd import get_log_likelihood
    from tests import fuzz_util
    _, _, model = fuzz_util.build_model()
    g = np.random.Generator(np.random.PCG64(31))
    prompt = torch.from_numpy(g.integers(0, fuzz_util.MASK_ID, size=21, dtype=np.int64))
    answer = torch.from_numpy(g.integers(0, fuzz_util.MASK_ID, size=13, dtype=np.int64))
    for cfg, extra in ((0.0, {}), (1.5, {}), (0.0, dict(rank=1, world_size=3))):
        kw = dict(mc_num=24, batch_size=6, cfg_scale=cfg, mask_id=fuzz_util.MASK_ID, seed=5, return_loss_sums=True, **extra)
        loop = get_log_likelihood(model, prompt, answer, fuse_iterations=False, **kw).cpu().numpy()
        fused = get_log_likelihood(model, prompt, answer, fuse_iterations=True, **kw).cpu().numpy()
        assert loop.shape == fused.shape == (4,) and (loop > 0).all()
        assert np.abs(fused - loop).max() <= 2e-3 * np.abs(loop).max(), (cfg, extra, fused, loop)
    a = get_log_likelihood(model, prompt, answer, mc_num=24, batch_size=6, mask_id=fuzz_util.MASK_ID, seed=5)
    b = get_log_likelihood(model, prompt, answer, mc_num=24, batch_size=6, mask_id=fuzz_util.MASK_ID, seed=5, fuse_iterations=False)
    assert abs(a - b) <= 1e-3 * abs(b)
    # injected masks (k_fn / perm_fn) take the same route
    ks, perms = [3, 9, 1, 13], [g.permutation(13) for _ in range(24)]
    outs = []
    for fuse in (True, False):
        k1, p1 = list(ks), iter(perms)
        outs.append(get_log_likelihood(model, prompt, answer, mc_num=24, batch_size=6, mask_id=fuzz_util.MASK_ID,
                                       k_fn=lambda T: k1.pop(0), perm_fn=lambda T: next(p1), fuse_iterations=fuse))
    assert abs(outs[0] - outs[1]) <= 1e-3 * abs(outs[1])
