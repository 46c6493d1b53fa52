"""Pins the CPU oracle against the golden vectors captured from the reference itself
(tools/make_golden.py ran /root/reference/generate.py and get_log_likelihood.py)."""
import numpy as np
import pytest

from oracle import likelihood as oll
from oracle import sampler as osamp
from oracle.bf16 import ulp_distance
from tests.golden_util import Draws, FakeModel, MASK_ID, case_confidence, fake_model_logits, load


def _cases():
    _, meta = load("sampler.npz")
    return meta


@pytest.fixture(scope="module")
def sampler_z():
    return load("sampler.npz")[0]


def _step_logits(case, call, B, L):
    Be = 2 * B if case["cfg"] > 0 else B
    lg = fake_model_logits(case["seed"], call, (Be, L, case["V"]), case.get("boost_ids", ()), case.get("boost", 0.0))
    if case["cfg"] > 0:
        return osamp.cfg_combine(lg[:B], lg[B:], case["cfg"])
    return lg


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_sampler_teacher_forced(case, sampler_z):
    """Every step of the reference run: given the reference's x_in, the oracle's confidences
    match (<= 1 bf16 ulp; exact for 'random'), the reference's transfer set is a valid top-k of
    the oracle's confidences, and the transferred tokens are the oracle's candidates."""
    z = sampler_z
    n = case["name"]
    xs = z[f"{n}/x_steps"].astype(np.int64)
    ks = z[f"{n}/k"]
    conf_ref = case_confidence(z, n)
    B, L = xs.shape[1], xs.shape[2]
    P, bl = case["P"], case["bl"]
    steps_per_block = case["steps"] // (case["G"] // bl)
    draws = Draws(case["seed"] + 5000)
    for t in range(xs.shape[0] - 1):
        nb = t // steps_per_block
        logits = _step_logits(case, t, B, L)
        u = draws.rand_like(logits.shape) if case["T"] != 0 else None
        rs = draws.rand((B, L)) if case["remasking"] == "random" else None
        x0, conf = osamp.confidence_step(xs[t], logits, MASK_ID, P + (nb + 1) * bl, case["remasking"], case["T"], u,
                                         rs, case["eos_inf"], case["conf_inf"], torch_cpu=True)
        cr = conf_ref[t]
        assert np.array_equal(np.isneginf(conf), np.isneginf(cr))
        assert np.array_equal(np.isnan(conf), np.isnan(cr))
        fin = np.isfinite(cr)
        if case["remasking"] == "random":
            assert np.array_equal(conf[fin], cr[fin])
        else:
            assert ulp_distance(conf[fin], cr[fin]).max(initial=0) <= 1
        changed = xs[t + 1] != xs[t]
        for j in range(B):
            assert osamp.is_valid_topk(cr[j], changed[j], ks[t, j])
        assert np.array_equal(xs[t + 1][changed], x0[changed])


def _boundary_tie_free(case, z):
    n = case["name"]
    conf = case_confidence(z, n)
    ks = z[f"{n}/k"]
    for t in range(conf.shape[0]):
        for j in range(conf.shape[1]):
            k = int(ks[t, j])
            row = conf[t, j]
            if np.isnan(row).any():
                return False
            s = np.sort(row)[::-1]
            if 0 < k < len(s) and s[k - 1] == s[k]:
                return False
    return True


def test_sampler_free_running(sampler_z):
    """Whole generate() runs: wherever the reference run had no tie at a top-k boundary the
    oracle reproduces every intermediate x exactly (ties are resolved by torch.topk in an
    unspecified order, so those cases are covered by the teacher-forced test only)."""
    z = sampler_z
    exact = total = 0
    for case in _cases():
        n = case["name"]
        if not _boundary_tie_free(case, z):
            continue
        total += 1
        model = FakeModel(case["seed"], case["V"], case.get("boost_ids", ()), case.get("boost", 0.0))
        draws = Draws(case["seed"] + 5000)
        am = z[f"{n}/attention_mask"] if f"{n}/attention_mask" in z.files else None
        trace = []
        out = osamp.generate(model, z[f"{n}/prompt"], am, steps=case["steps"], gen_length=case["G"],
                             block_length=case["bl"], temperature=case["T"], cfg_scale=case["cfg"],
                             remasking=case["remasking"], mask_id=MASK_ID, logits_eos_inf=case["eos_inf"],
                             confidence_eos_eot_inf=case["conf_inf"], noise_fn=draws.rand_like, rand_fn=draws.rand,
                             trace=trace, torch_cpu=True)
        xs = z[f"{n}/x_steps"].astype(np.int64)
        ok = np.array_equal(out, z[f"{n}/final"]) and all(
            np.array_equal(tr["x_in"], xs[i]) for i, tr in enumerate(trace))
        exact += ok
    assert total >= 40
    # a 1-ulp difference in a bf16 confidence (different float32 summation order) can flip a
    # near-tie; allow a handful of such cases
    assert exact >= total - 3, (exact, total)


def test_sampler_invariants(sampler_z):
    """The two invariants visible in /root/reference/visualization/sample_process.txt: step t
    unmasks exactly k tokens per row and revealed tokens never change."""
    z = sampler_z
    for case in _cases():
        n = case["name"]
        xs = z[f"{n}/x_steps"].astype(np.int64)
        ks = z[f"{n}/k"]
        for t in range(xs.shape[0] - 1):
            was = xs[t] == MASK_ID
            now = xs[t + 1] == MASK_ID
            assert np.array_equal((was & ~now).sum(axis=1), ks[t])
            assert np.array_equal(xs[t + 1][~was], xs[t][~was])
        assert not (xs[-1] == MASK_ID).any()


def test_cfg_without_mask_is_the_reference_error():
    with pytest.raises(UnboundLocalError):
        osamp.generate(FakeModel(1, 64), np.zeros((1, 4), dtype=np.int64), None, steps=2, gen_length=4,
                       block_length=4, cfg_scale=1.0)


def test_unknown_remasking():
    with pytest.raises(NotImplementedError):
        osamp.generate(FakeModel(1, 64), np.zeros((1, 4), dtype=np.int64), None, steps=2, gen_length=4,
                       block_length=4, remasking="nope")


def test_num_transfer_tokens():
    m = np.zeros((3, 10), dtype=bool)
    m[0, :7] = True
    m[1, :] = True
    out = osamp.get_num_transfer_tokens(m, 4)
    assert out.tolist() == [[2, 2, 2, 1], [3, 3, 2, 2], [0, 0, 0, 0]]


def test_linspace_table():
    z, _ = load("linspace.npz")
    off = 0
    for T, b, k in zip(z["T"], z["b"], z["k"]):
        got = oll.stratified_counts(int(k), int(b), int(T))
        assert np.array_equal(got, z["x"][off: off + b]), (T, b, k)
        off += b


def _ll_cases():
    return load("likelihood.npz")[1]


@pytest.mark.parametrize("case", _ll_cases(), ids=lambda c: c["name"])
def test_likelihood(case):
    z, _ = load("likelihood.npz")
    n = case["name"]
    ks = list(z[f"{n}/k"])
    perms = z[f"{n}/perms"].reshape(-1, case["l2"])
    it = iter(perms)
    model = FakeModel(case["seed"], case["V"])
    trace = []
    val = oll.get_log_likelihood(model, z[f"{n}/prompt"], z[f"{n}/answer"], mc_num=case["mc"],
                                 batch_size=case["b"], cfg_scale=case["cfg"], mask_id=MASK_ID,
                                 k_fn=lambda T: int(ks.pop(0)), perm_fn=lambda T: next(it).astype(np.int64),
                                 trace=trace, torch_cpu=True)
    for i, tr in enumerate(trace):
        assert np.array_equal(tr["perturbed"], z[f"{n}/perturbed"][i])
    ref = float(z[f"{n}/value"])
    assert abs(val - ref) <= 1e-3 * abs(ref), (val, ref)
    # bf16-rounded per-token CE makes the two agree far more tightly than the contract asks
    assert abs(val - ref) <= 2e-5 * abs(ref), (val, ref)


def _f32lse_cases():
    return load("sampler_f32lse.npz")[1]


@pytest.mark.parametrize("case", _f32lse_cases(), ids=lambda c: c["name"])
def test_entropy_rules_float32_softmax_runs(case):
    """The entropy rules against reference runs whose F.softmax / F.log_softmax computed in float32 and rounded
    once to bf16 (what torch's GPU kernels do for a bf16 tensor; tools/make_golden.py::Float32Softmax): the oracle's
    DEFAULT branch (torch_cpu=False, the form the HIP kernel implements) reproduces every step's scores to one
    bf16 ulp, the reference's transfers are a valid top-k of them, and the committed tokens are its candidates.
    (Entropies below 2^-16 are compared absolutely, to 2^-22: when one token holds all the mass its term is
    log(1 + eps) with eps near float32's resolution at 1, so the last bits depend on the order in which the row's
    float32 sum was taken -- a 2-ulp difference of a 8.5e-6 entropy is the one such value in these runs.)"""
    z, _ = load("sampler_f32lse.npz")
    n = case["name"]
    xs = z[f"{n}/x_steps"].astype(np.int64)
    ks = z[f"{n}/k"]
    conf_ref = case_confidence(z, n)
    B, L = xs.shape[1], xs.shape[2]
    P, bl = case["P"], case["bl"]
    steps_per_block = case["steps"] // (case["G"] // bl)
    draws = Draws(case["seed"] + 5000)
    worst = 0
    for t in range(xs.shape[0] - 1):
        nb = t // steps_per_block
        logits = _step_logits(case, t, B, L)
        u = draws.rand_like(logits.shape) if case["T"] != 0 else None
        x0, conf = osamp.confidence_step(xs[t], logits, MASK_ID, P + (nb + 1) * bl, case["remasking"], case["T"], u,
                                         None, case["eos_inf"], case["conf_inf"])
        cr = conf_ref[t]
        assert np.array_equal(np.isneginf(conf), np.isneginf(cr)) and np.array_equal(np.isnan(conf), np.isnan(cr))
        fin = np.isfinite(cr)
        tiny = fin & (np.abs(cr) < 2.0 ** -16)
        assert np.abs(conf[tiny] - cr[tiny]).max(initial=0) <= 2.0 ** -22
        fin &= ~tiny
        worst = max(worst, int(ulp_distance(conf[fin], cr[fin]).max(initial=0)))
        changed = xs[t + 1] != xs[t]
        for j in range(B):
            assert osamp.is_valid_topk(cr[j], changed[j], ks[t, j])
        assert np.array_equal(xs[t + 1][changed], x0[changed])
    assert worst <= 1, worst
