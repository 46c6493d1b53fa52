"""Parity of the remasking epilogue and of generate() (through the C ABI / the Python drop-in)
against the oracle and the golden vectors captured from the reference, on a real MI355X."""
import ctypes as C

import numpy as np
import pytest

from oracle import rng as orng
from oracle import sampler as osamp
from oracle.bf16 import ulp_distance
from tests.golden_util import Draws, MASK_ID, case_confidence, fake_model_logits, load

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EOS_ID, EOT_ID = 126081, 126348


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(lib, rc):
    assert rc == 0, lib.llada_last_error().decode()


ROW_MODE = {"splits": None}


@pytest.fixture(autouse=True, params=[None, 1, 3], ids=["whole_row", "stream_1", "stream_3"])
def row_mode(request, lib):
    """Every test runs three times: the whole-row kernels (no workspace), and the streamed row pieces
    (llada_rows_workspace_bytes) with one and with three pieces per row."""
    ROW_MODE["splits"] = request.param
    lib.llada_rows_force_splits(request.param or 0)
    yield
    lib.llada_rows_force_splits(0)
    ROW_MODE["splits"] = None


def remask(lib, cond, un, cfg, x, block_end, remasking, flags, T, noise=None, rand_scores=None, seed=0, step=0,
           row_offset=0, row_pos=None):
    from llada_amd import _lib
    B, L = x.shape
    V = cond.shape[-1]
    xt = dev(x)
    x0 = torch.full((B, L), -7, dtype=torch.long, device="cuda")
    conf = torch.full((B, L), 123.0, dtype=torch.float32, device="cuda")
    ct = dev(cond.astype(np.float32), torch.bfloat16)
    ut = None if un is None else dev(un.astype(np.float32), torch.bfloat16)
    nt = None if noise is None else dev(noise.astype(np.float64))
    rt = None if rand_scores is None else dev(rand_scores.astype(np.float32))
    rp = None if row_pos is None else dev(row_pos.astype(np.int32))
    args = _lib.RemaskArgs(ct.data_ptr(), _lib.ptr(ut), float(cfg), _lib.ptr(rp), ct.shape[0], V, xt.data_ptr(), B, L,
                           block_end, MASK_ID, _lib.REMASK[remasking], flags, EOS_ID, EOT_ID, float(T), _lib.ptr(nt),
                           _lib.ptr(rt), seed, step, row_offset, x0.data_ptr(), conf.data_ptr())
    if ROW_MODE["splits"] is not None:
        ws = torch.empty(lib.llada_rows_workspace_bytes(ct.shape[0]), dtype=torch.uint8, device="cuda")
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib, lib.llada_remask_scores(C.byref(args), stream()))
    torch.cuda.synchronize()
    return x0.cpu().numpy(), conf.cpu().numpy()


def assert_x0(x0, ox0, oconf, x):
    """Candidate tokens agree wherever they can be used (confidence > -inf); elsewhere the
    kernel keeps x0 = x (the reference computes an argmax there that nothing reads)."""
    live = ~np.isneginf(oconf)
    assert np.array_equal(x0[live], ox0[live])
    assert np.array_equal(x0[~live], x[~live])


def compare_conf(got, ref, exact, abs_tol=0.0):
    """-inf / NaN patterns identical; finite values within 1 bf16 ulp (or abs_tol: the entropy of a
    nearly one-hot row is ~log(1 + eps) of a float32 sum and inherits that sum's last-bit noise)."""
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref)
    if exact:
        assert np.array_equal(got[fin], ref[fin])
    else:
        bad = (ulp_distance(got[fin], ref[fin]) > 1) & (np.abs(got[fin] - ref[fin]) > abs_tol)
        assert not bad.any()


def _cases():
    return load("sampler.npz")[1]


@pytest.fixture(scope="module")
def sampler_z():
    return load("sampler.npz")[0]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_remask_step_teacher_forced(lib, case, sampler_z):
    """Every step of every reference run: HIP epilogue == oracle (x0 exact, confidence <= 1 bf16
    ulp), == the reference's recorded confidence, and the top-k transfer reproduces a valid
    reference transfer set with identical confidences."""
    z = sampler_z
    n = case["name"]
    xs = z[f"{n}/x_steps"].astype(np.int64)
    ks = z[f"{n}/k"]
    conf_ref = case_confidence(z, n)
    B, L = xs.shape[1], xs.shape[2]
    P, bl, V = case["P"], case["bl"], case["V"]
    spb = case["steps"] // (case["G"] // bl)
    draws = Draws(case["seed"] + 5000)
    flags = (1 if case["eos_inf"] else 0) | (2 if case["conf_inf"] else 0)
    entropy = case["remasking"] in ("low_entropy", "high_entropy")
    for t in range(xs.shape[0] - 1):
        nb = t // spb
        be = P + (nb + 1) * bl
        Be = 2 * B if case["cfg"] > 0 else B
        lg = fake_model_logits(case["seed"], t, (Be, L, V), case.get("boost_ids", ()), case.get("boost", 0.0))
        cond = lg[:B].reshape(B * L, V)
        un = lg[B:].reshape(B * L, V) if case["cfg"] > 0 else None
        u = draws.rand_like((B, L, V)) if case["T"] != 0 else None
        rs = draws.rand((B, L)) if case["remasking"] == "random" else None
        x0, conf = remask(lib, cond, un, case["cfg"], xs[t], be, case["remasking"], flags, case["T"],
                          None if u is None else u.reshape(B * L, V), rs)
        logits = osamp.cfg_combine(lg[:B], lg[B:], case["cfg"]) if case["cfg"] > 0 else lg.copy()
        ox0, oconf = osamp.confidence_step(xs[t], logits, MASK_ID, be, case["remasking"], case["T"], u, rs,
                                           case["eos_inf"], case["conf_inf"])
        assert_x0(x0, ox0, oconf, xs[t])
        compare_conf(conf, oconf, exact=case["remasking"] == "random", abs_tol=4e-7 if entropy else 0.0)
        if not entropy:
            compare_conf(conf, conf_ref[t], exact=case["remasking"] == "random")
        else:
            # The reference's recorded entropy scores were computed on the CPU, where torch's bf16 log_softmax keeps
            # the row's log-sum-exp in a bf16 scratch; the kernel keeps it in float32, as a GPU run of the reference
            # does.  That shifts every log-probability by at most half a bf16 step of the log-sum-exp (<= log V =
            # 6.2 here: 2^-6), and the entropy by as much, before its own rounding to bf16 (another 2^-6 below 8):
            # bounded deviation instead of equality, same -inf / NaN pattern.
            ref_c = conf_ref[t].astype(np.float32)
            assert np.array_equal(np.isneginf(conf), np.isneginf(ref_c)) and np.array_equal(np.isnan(conf), np.isnan(ref_c))
            fin = np.isfinite(ref_c)
            assert np.abs(conf[fin] - ref_c[fin]).max(initial=0.0) <= 2.0 ** -5
        # top-k transfer on the reference's own confidence tensor
        xt = dev(xs[t].copy())
        sched = dev(ks[t].astype(np.int32).reshape(B, 1))
        x0t, cft = dev(ox0), dev(conf_ref[t].astype(np.float32))
        check(lib, lib.llada_topk_transfer(xt.data_ptr(), x0t.data_ptr(), cft.data_ptr(), B, L, sched.data_ptr(), 1, 0,
                                           stream()))
        torch.cuda.synchronize()
        x_new = xt.cpu().numpy()
        want = xs[t].copy()
        osamp.topk_transfer(want, ox0, conf_ref[t], ks[t])
        assert np.array_equal(x_new, want)                      # bit-exact vs the oracle's tie rule
        changed = x_new != xs[t]
        for j in range(B):
            assert osamp.is_valid_topk(conf_ref[t][j], changed[j], ks[t, j])
        ref_changed = xs[t + 1] != xs[t]
        tie_free = all(_no_boundary_tie(conf_ref[t][j], ks[t, j]) for j in range(B))
        if tie_free:
            assert np.array_equal(changed, ref_changed)         # identical to the reference's transfer set


def _f32lse_cases():
    return load("sampler_f32lse.npz")[1]


@pytest.mark.parametrize("case", _f32lse_cases(), ids=lambda c: c["name"])
def test_entropy_rules_against_float32_softmax_reference_runs(lib, case):
    """The two entropy rules pinned to the reference: runs of generate.py whose F.softmax / F.log_softmax computed in
    float32 and rounded once to bf16 (torch's GPU arithmetic for a bf16 tensor; tests/golden/sampler_f32lse.npz,
    tools/make_golden.py::Float32Softmax).  Every step: the kernel's scores equal the recorded ones to one bf16 ulp
    (entropies below 2^-16 absolutely, to 2^-22: log(1 + eps) of a float32 sum), candidates are the oracle's, and the
    top-k transfer on the kernel's own scores is a valid reference transfer set wherever the boundary has no tie."""
    z, _ = load("sampler_f32lse.npz")
    n = case["name"]
    xs = z[f"{n}/x_steps"].astype(np.int64)
    ks = z[f"{n}/k"]
    conf_ref = case_confidence(z, n)
    B, L = xs.shape[1], xs.shape[2]
    P, bl, V = case["P"], case["bl"], case["V"]
    spb = case["steps"] // (case["G"] // bl)
    draws = Draws(case["seed"] + 5000)
    for t in range(xs.shape[0] - 1):
        be = P + (t // spb + 1) * bl
        Be = 2 * B if case["cfg"] > 0 else B
        lg = fake_model_logits(case["seed"], t, (Be, L, V))
        u = draws.rand_like((B, L, V)) if case["T"] != 0 else None
        x0, conf = remask(lib, lg[:B].reshape(B * L, V), lg[B:].reshape(B * L, V) if case["cfg"] > 0 else None, case["cfg"],
                          xs[t], be, case["remasking"], 0, case["T"], None if u is None else u.reshape(B * L, V))
        cr = conf_ref[t].astype(np.float32)
        assert np.array_equal(np.isneginf(conf), np.isneginf(cr)) and np.array_equal(np.isnan(conf), np.isnan(cr))
        fin = np.isfinite(cr)
        tiny = fin & (np.abs(cr) < 2.0 ** -16)
        assert np.abs(conf[tiny] - cr[tiny]).max(initial=0.0) <= 2.0 ** -22
        assert ulp_distance(conf[fin & ~tiny], cr[fin & ~tiny]).max(initial=0) <= 1
        changed = xs[t + 1] != xs[t]
        assert np.array_equal(x0[changed], xs[t + 1][changed])                   # the committed tokens are its candidates
        if all(_no_boundary_tie(cr[j], ks[t, j]) and _no_boundary_tie(conf[j], ks[t, j]) for j in range(B)) \
                and ulp_distance(conf[fin], cr[fin]).max(initial=0) == 0:
            want = xs[t].copy()
            osamp.topk_transfer(want, x0, conf, ks[t])
            assert np.array_equal(want, xs[t + 1])


def _no_boundary_tie(row, k):
    k = int(k)
    if np.isnan(row).any():
        return False
    s = np.sort(row)[::-1]
    return not (0 < k < len(s) and s[k - 1] == s[k])


def test_remask_rows_subset_and_defaults(lib):
    """Active-row mode: logits only for the current block; everything else gets the defaults."""
    B, L, V, P, bl = 3, 40, 512, 8, 16
    g = np.random.Generator(np.random.PCG64(1))
    x = g.integers(0, 400, size=(B, L), dtype=np.int64)
    x[:, P:] = MASK_ID
    x[1, P + 3] = 17                       # already revealed inside the block
    rows = (np.arange(B)[:, None] * L + np.arange(P, P + bl)[None, :]).reshape(-1)
    lg_full = fake_model_logits(3, 0, (B, L, V))
    x0, conf = remask(lib, lg_full.reshape(B * L, V)[rows], None, 0.0, x, P + bl, "low_confidence", 0, 0.0, row_pos=rows)
    ox0, oconf = osamp.confidence_step(x, lg_full.copy(), MASK_ID, P + bl, "low_confidence")
    assert_x0(x0, ox0, oconf, x)
    compare_conf(conf, oconf, exact=False)
    assert np.isneginf(conf[:, P + bl:]).all() and np.isneginf(conf[:, :P]).all() and np.isneginf(conf[1, P + 3])


@pytest.mark.parametrize("remasking", ["low_confidence", "random"])
def test_remask_philox_streams(lib, remasking):
    """Temperature > 0 without injected noise: the device Philox stream == its numpy twin."""
    B, L, V, P = 2, 24, 512, 8
    seed, step, row_offset = 0x1234ABCD5678, 5, 3
    g = np.random.Generator(np.random.PCG64(2))
    x = g.integers(0, 400, size=(B, L), dtype=np.int64)
    x[:, P:] = MASK_ID
    lg = fake_model_logits(4, 0, (B, L, V))
    x0, conf = remask(lib, lg.reshape(B * L, V), None, 0.0, x, L, remasking, 0, 0.7, seed=seed, step=step,
                      row_offset=row_offset)
    u = orng.gumbel_uniforms(seed, step, row_offset + np.arange(B), np.arange(L), V)
    rs = orng.remask_uniforms(seed, step, row_offset + np.arange(B), np.arange(L))
    ox0, oconf = osamp.confidence_step(x, lg.copy(), MASK_ID, L, remasking, 0.7, u, rs)
    assert_x0(x0, ox0, oconf, x)
    compare_conf(conf, oconf, exact=remasking == "random")


def test_remask_full_vocab(lib):
    """One full-vocabulary (126464) step with CFG and both EOS controls."""
    B, L, V, P = 1, 6, 126464, 2
    x = np.array([[5, 9, MASK_ID, MASK_ID, 11, MASK_ID]], dtype=np.int64)
    lg = fake_model_logits(77, 0, (2 * B, L, V), (EOS_ID, EOT_ID), 9.0)
    for T in (0.0, 0.7):
        u = np.random.Generator(np.random.PCG64(5)).random((B, L, V)) if T else None
        x0, conf = remask(lib, lg[:B].reshape(B * L, V), lg[B:].reshape(B * L, V), 1.0, x, L, "low_confidence", 3, T,
                          None if u is None else u.reshape(B * L, V))
        logits = osamp.cfg_combine(lg[:B], lg[B:], 1.0)
        ox0, oconf = osamp.confidence_step(x, logits, MASK_ID, L, "low_confidence", T, u, None, True, True)
        assert_x0(x0, ox0, oconf, x)
        compare_conf(conf, oconf, exact=False)


@pytest.mark.parametrize("remasking,flags", [(r, f) for r in ("low_confidence", "high_confidence", "random") for f in (0, 1, 2, 3)]
                         + [("low_entropy", 0), ("high_entropy", 0), ("low_entropy", 1), ("high_entropy", 3)])
def test_remask_full_vocab_planted_rows(lib, remasking, flags):
    """Full-vocabulary rows built to hit the seams of the row kernels: the maximum repeated in far-apart
    places (the first index must win whichever piece finds it), the maximum on EOS / on EoT (removed
    before the arg-max, or only from the softmax, depending on the flags), the maximum in the very first
    and the very last entry, a row that is -inf except for one entry, and an inactive row.  Runs on the
    whole-row kernel, on 1 and 3 forced pieces, and on the automatic piece count."""
    V, L = 126464, 9
    g = np.random.Generator(np.random.PCG64(11))
    lg = (g.standard_normal((1, L, V), dtype=np.float32) * 2.0).astype(np.float32)
    top = 30.0
    lg[0, 0, [70000, 123, 126000]] = top                 # repeated maximum: 123 wins
    lg[0, 1, [EOS_ID, 90001]] = [top, top - 1.0]         # maximum on EOS
    lg[0, 2, [EOT_ID, 15]] = [top, top - 0.5]            # maximum on EoT
    lg[0, 3, 0] = top
    lg[0, 4, V - 1] = top
    lg[0, 5, :] = -np.inf
    lg[0, 5, 77777] = 1.0
    lg[0, 6, [EOS_ID, EOT_ID]] = top                     # both special tokens tie for the maximum
    lg[0, 7, [EOT_ID + 8, EOS_ID - 8]] = top             # neighbours of the special tokens
    from oracle.bf16 import round_bf16
    lg = round_bf16(lg)
    x = np.full((1, L), MASK_ID, dtype=np.int64)
    x[0, 8] = 42                                          # inactive row
    rs = g.random((1, L)).astype(np.float32)
    modes = [ROW_MODE["splits"]] + ([0] if ROW_MODE["splits"] == 1 else [])
    try:
        for mode in modes:
            ROW_MODE["splits"] = mode
            lib.llada_rows_force_splits(mode or 0)
            x0, conf = remask(lib, lg.reshape(L, V), None, 0.0, x, L, remasking, flags, 0.0, rand_scores=rs)
            ox0, oconf = osamp.confidence_step(x, lg.copy(), MASK_ID, L, remasking, 0.0, None, rs, bool(flags & 1),
                                               bool(flags & 2))
            assert_x0(x0, ox0, oconf, x)
            # entropy rules: a -inf logit makes the reference's score NaN (generate.py:113,118: -inf * 0), which
            # torch.topk ranks highest; finite scores are float32 sums of 126464 bf16 terms (summation order)
            compare_conf(conf, oconf, exact=remasking == "random", abs_tol=2e-6 if "entropy" in remasking else 0.0)
    finally:
        ROW_MODE["splits"] = modes[0]



def test_add_gumbel_noise_helper(lib):
    """llada_amd.generate.add_gumbel_noise (generate.py:8-19 as a stand-alone map): temperature 0 returns the
    same object; with injected uniforms it equals the oracle's float64 formula; with the library's Philox stream
    its arg-max is the candidate token llada_remask_scores picks for the same (seed, step, row_offset)."""
    from llada_amd.generate import add_gumbel_noise
    B, L, V = 2, 5, 512
    lg = fake_model_logits(21, 0, (B, L, V))
    t = dev(lg, torch.bfloat16).reshape(B, L, V)
    assert add_gumbel_noise(t, 0) is t
    u = np.random.Generator(np.random.PCG64(8)).random((B, L, V))
    got = add_gumbel_noise(t, 0.7, noise=torch.from_numpy(u)).cpu().numpy()
    want = osamp.add_gumbel_noise(lg, 0.7, u)
    assert got.dtype == np.float64 and np.allclose(got, want, rtol=1e-12, atol=0.0)
    seed, step, row_offset = 77, 3, 5
    noisy = add_gumbel_noise(t, 0.7, seed=seed, step=step, row_offset=row_offset)
    x = np.full((B, L), MASK_ID, dtype=np.int64)
    x0, _ = remask(lib, lg.reshape(B * L, V), None, 0.0, x, L, "low_confidence", 0, 0.7, seed=seed, step=step,
                   row_offset=row_offset)
    assert np.array_equal(noisy.argmax(-1).cpu().numpy(), x0)


def test_num_transfer_tokens(lib):
    B, L = 4, 50
    g = np.random.Generator(np.random.PCG64(6))
    x = g.integers(0, 100, size=(B, L), dtype=np.int64)
    x[0, 10:30] = MASK_ID
    x[1, 12:19] = MASK_ID
    x[3, 5:45] = MASK_ID
    for steps in (1, 3, 8, 20):
        sched = torch.empty((B, steps), dtype=torch.int32, device="cuda")
        xt = dev(x)
        check(lib, lib.llada_num_transfer_tokens(xt.data_ptr(), B, L, 10, 30, MASK_ID, steps, sched.data_ptr(), stream()))
        torch.cuda.synchronize()
        ref = osamp.get_num_transfer_tokens(x[:, 10:30] == MASK_ID, steps)
        assert np.array_equal(sched.cpu().numpy(), ref)


def test_topk_edge_cases(lib):
    """k = 0, k = all candidates, NaN ranks highest, -0 == +0, k beyond the candidates."""
    L = 300
    conf = np.full((5, L), -np.inf, dtype=np.float32)
    g = np.random.Generator(np.random.PCG64(7))
    conf[0, 40:72] = g.random(32, dtype=np.float32)
    conf[1, 40:72] = 0.5                                  # all tied
    conf[2, 40:72] = g.random(32, dtype=np.float32)
    conf[2, 50] = np.nan
    conf[3, 40:72] = 0.0
    conf[3, 60] = -0.0
    conf[4, 10:14] = 1.0
    ks = np.array([0, 5, 1, 31, 6], dtype=np.int32)       # row 4: k exceeds the 4 candidates
    x = np.zeros((5, L), dtype=np.int64)
    x0 = np.arange(5 * L, dtype=np.int64).reshape(5, L) + 1
    xt = dev(x.copy())
    x0t, cft, kst = dev(x0), dev(conf), dev(ks.reshape(5, 1))
    check(lib, lib.llada_topk_transfer(xt.data_ptr(), x0t.data_ptr(), cft.data_ptr(), 5, L, kst.data_ptr(), 1, 0, stream()))
    torch.cuda.synchronize()
    got = xt.cpu().numpy()
    want = x.copy()
    osamp.topk_transfer(want, x0, conf, ks)
    assert np.array_equal(got, want)
    assert (got[0] == 0).all() and got[2, 50] == x0[2, 50] and (got[1, 40:45] != 0).all()


class TorchFakeModel:
    """A foreign model that only honours the reference's protocol (generate.py:60,88,92)."""

    def __init__(self, seed, vocab, boost_ids=(), boost=0.0):
        self.device = torch.device("cuda")
        self.seed, self.vocab, self.calls = seed, vocab, 0
        self.boost_ids, self.boost = tuple(boost_ids), boost

    def __call__(self, x, attention_mask=None):
        lg = fake_model_logits(self.seed, self.calls, tuple(x.shape) + (self.vocab,), self.boost_ids, self.boost)
        self.calls += 1

        class Out:
            pass
        o = Out()
        o.logits = dev(lg, torch.bfloat16)
        return o


def test_generate_matches_reference_runs(lib, sampler_z):
    """llada_amd.generate() end to end (foreign-model path) reproduces the reference's final
    sequences wherever the reference run had no tie at a top-k boundary."""
    from llada_amd import generate
    z = sampler_z
    exact = total = 0
    for case in _cases():
        n = case["name"]
        conf = case_confidence(z, n)
        ks = z[f"{n}/k"]
        if case["remasking"] in ("low_entropy", "high_entropy"):
            continue      # these runs went through torch's CPU-only bf16 log-sum-exp; see the float32-softmax runs below
        if not all(_no_boundary_tie(conf[t, j], ks[t, j]) for t in range(conf.shape[0]) for j in range(conf.shape[1])):
            continue
        total += 1
        model = TorchFakeModel(case["seed"], case["V"], case.get("boost_ids", ()), case.get("boost", 0.0))
        draws = Draws(case["seed"] + 5000)
        am = torch.from_numpy(z[f"{n}/attention_mask"]) if f"{n}/attention_mask" in z.files else None
        out = generate(model, torch.from_numpy(z[f"{n}/prompt"]), am, steps=case["steps"], gen_length=case["G"],
                       block_length=case["bl"], temperature=case["T"], cfg_scale=case["cfg"],
                       remasking=case["remasking"], mask_id=MASK_ID, logits_eos_inf=case["eos_inf"],
                       confidence_eos_eot_inf=case["conf_inf"],
                       noise_fn=lambda s: torch.from_numpy(draws.rand_like(s)),
                       rand_fn=lambda s: torch.from_numpy(draws.rand(s)))
        assert out.dtype == torch.long and out.device.type == "cuda"
        exact += np.array_equal(out.cpu().numpy(), z[f"{n}/final"])
    assert total >= 30
    assert exact >= total - 2, (exact, total)    # a 1-ulp confidence difference may flip a near tie


def test_generate_matches_float32_softmax_entropy_runs(lib):
    """llada_amd.generate() end to end with the entropy rules against the reference runs whose softmax pair computed
    in float32 (sampler_f32lse.npz): the final sequences are the reference's wherever no step had a tie at its top-k
    boundary (a 1-ulp score difference may still flip a near tie: a handful of runs are allowed to differ)."""
    from llada_amd import generate
    z, meta = load("sampler_f32lse.npz")
    exact = total = 0
    for case in meta:
        n = case["name"]
        conf = case_confidence(z, n)
        ks = z[f"{n}/k"]
        if not all(_no_boundary_tie(conf[t, j], ks[t, j]) for t in range(conf.shape[0]) for j in range(conf.shape[1])):
            continue
        total += 1
        model = TorchFakeModel(case["seed"], case["V"])
        draws = Draws(case["seed"] + 5000)
        am = torch.from_numpy(z[f"{n}/attention_mask"]) if f"{n}/attention_mask" in z.files else None
        out = generate(model, torch.from_numpy(z[f"{n}/prompt"]), am, steps=case["steps"], gen_length=case["G"],
                       block_length=case["bl"], temperature=case["T"], cfg_scale=case["cfg"], remasking=case["remasking"],
                       mask_id=MASK_ID, noise_fn=lambda s: torch.from_numpy(draws.rand_like(s)))
        exact += np.array_equal(out.cpu().numpy(), z[f"{n}/final"])
    assert total >= 20, total
    assert exact >= total - 2, (exact, total)


def test_generate_errors(lib):
    from llada_amd import generate
    m = TorchFakeModel(1, 512)
    p = torch.zeros((1, 4), dtype=torch.long)
    with pytest.raises(AssertionError):
        generate(m, p, steps=4, gen_length=6, block_length=4)
    with pytest.raises(AssertionError):
        generate(m, p, steps=3, gen_length=8, block_length=4)
    with pytest.raises(NotImplementedError):
        generate(m, p, steps=2, gen_length=4, block_length=4, remasking="nope")
    with pytest.raises(UnboundLocalError):
        generate(m, p, steps=2, gen_length=4, block_length=4, cfg_scale=1.0)
