"""suffix_greedy_prediction and the evaluation-harness mirror on a real MI355X, against the
oracle's restatement (eval_llada.py:172-279).  Parity unpinned against the reference itself:
eval_llada.py needs lm_eval to import."""
import numpy as np
import pytest

from oracle import harness as oh
from oracle.model import LLaDAConfig as OCfg
from oracle.model import init_weights
from tests.golden_util import fake_model_logits, load
from tests.test_harness_cpu import CharTokenizer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class TorchFake:
    """A foreign model object with the reference's protocol, logits from the documented stream."""

    def __init__(self, seed, vocab):
        self.seed, self.vocab, self.calls, self.device = seed, vocab, 0, torch.device("cuda")

    def __call__(self, x, attention_mask=None):
        lg = fake_model_logits(self.seed, self.calls, tuple(x.shape) + (self.vocab,))
        self.calls += 1

        class Out:
            logits = torch.from_numpy(lg).to("cuda", torch.bfloat16)
        return Out()


class NumpyFake:
    def __init__(self, seed, vocab):
        self.seed, self.vocab, self.calls = seed, vocab, 0

    def __call__(self, x, attention_mask=None):
        lg = fake_model_logits(self.seed, self.calls, tuple(x.shape) + (self.vocab,))
        self.calls += 1
        return lg


@pytest.mark.parametrize("cfg", [0.0, 1.5])
@pytest.mark.parametrize("vocab,mask_id", [(1024, 1023), (126464, 126336)])
def test_suffix_greedy_foreign_model_matches_oracle(lib, cfg, vocab, mask_id):
    from llada_amd.eval_harness import suffix_greedy_prediction
    g = np.random.Generator(np.random.PCG64(3))
    T = 9 if vocab > 2000 else 17
    prefix = g.integers(0, 1000, size=6, dtype=np.int64)
    target = g.integers(0, 1000, size=T, dtype=np.int64)
    trace, otrace = [], []
    got = suffix_greedy_prediction(TorchFake(11, vocab), torch.from_numpy(prefix), torch.from_numpy(target), cfg,
                                   mask_id, trace=trace)
    ok, seq = oh.suffix_greedy_prediction(NumpyFake(11, vocab), prefix, target, cfg, mask_id, trace=otrace)
    assert len(trace) == len(otrace) == T
    for a, b in zip(trace, otrace):
        assert np.array_equal(a.cpu().numpy(), b)
    assert bool(got) == ok and got.dtype == torch.bool and got.dim() == 0


def build_tiny():
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    z, _ = load("forward_standin.npz")
    d, nl, nh, ff, V = (int(v) for v in z["cfg"])
    ocfg = OCfg(d_model=d, n_layers=nl, n_heads=nh, mlp_hidden=ff, vocab_size=V)
    W = init_weights(ocfg, seed=int(z["weight_seed"][0]), std=float(z["weight_std"][0]), head_scale=float(z["head_scale"][0]))
    cfg = LLaDAConfig(d_model=d, n_layers=nl, n_heads=nh, mlp_hidden=ff, vocab_size=V, rope_theta=ocfg.rope_theta,
                      rms_eps=ocfg.rms_eps)
    return LLaDAModel(LLaDAWeights.from_numpy(cfg, W, "cuda")), V


@pytest.mark.parametrize("cfg", [0.0, 0.5])
def test_suffix_greedy_hip_model_teacher_forced(lib, cfg):
    """On the HIP model (answer-row logits only) every step must equal the oracle's step on the HIP
    model's own full logits for the same input (the oracle model's logits differ by bf16 rounding,
    which flips near-ties of a random model)."""
    from llada_amd.eval_harness import suffix_greedy_prediction
    model, V = build_tiny()
    mask_id = V - 1
    g = np.random.Generator(np.random.PCG64(21))
    prefix = g.integers(0, mask_id, size=8, dtype=np.int64)
    target = g.integers(0, mask_id, size=12, dtype=np.int64)
    trace = []
    got = suffix_greedy_prediction(model, torch.from_numpy(prefix), torch.from_numpy(target), cfg, mask_id, trace=trace)
    seq = np.concatenate([prefix, np.full(12, mask_id)])[None, :]

    def one_call(x, am):
        return model(torch.from_numpy(x).cuda()).logits.float().cpu().numpy()
    for cur in trace:
        # one oracle step from the same state: run its loop for a single iteration
        l1 = len(prefix)
        mask_index = seq == mask_id
        lg = oh.get_logits(one_call, seq, np.arange(seq.shape[1]) < l1, cfg, mask_id)[mask_index]
        x0 = lg.argmax(-1)
        p = oh.softmax_f32(lg)
        conf = p[np.arange(len(x0)), x0]
        cur = cur.cpu().numpy()
        changed = np.flatnonzero(cur != seq[0])
        assert len(changed) == 1
        pos_in_masked = int(np.flatnonzero(np.flatnonzero(mask_index[0]) == changed[0])[0])
        assert cur[changed[0]] == x0[pos_in_masked]
        assert conf[pos_in_masked] >= conf.max() * (1 - 1e-5)     # the float32 maximum (last-bit sum order aside)
        seq = cur[None, :].copy()
    assert bool(got) == bool(np.array_equal(seq[0, 8:], target))


def test_harness_requests_end_to_end(lib):
    """loglikelihood / generate_until through the harness mirror with a toy tokenizer: shapes, types,
    stop-sequence trimming and agreement of the likelihood with the drop-in called directly."""
    from llada_amd import get_log_likelihood
    from llada_amd.eval_harness import LLaDAEvalHarness
    model, V = build_tiny()
    mask_id = V - 1

    class Tok(CharTokenizer):
        def __call__(self, text):
            return {"input_ids": [ord(c) % (V - 2) + 1 for c in text]}

        def decode(self, ids, skip_special_tokens=False):
            return "".join(chr(97 + int(i) % 26) for i in ids)

    class Req:
        def __init__(self, *args):
            self.args = args

    h = LLaDAEvalHarness(model=model, tokenizer=Tok(), mask_id=mask_id, batch_size=4, mc_num=8, is_check_greedy=True,
                         steps=8, gen_length=16, block_length=8, seed=77)
    reqs = [Req("The sky is ", "blue"), Req("2+2=", " 4")]
    out = h.loglikelihood(reqs)
    assert len(out) == 2 and all(isinstance(ll, float) and ll < 0 and g in (0.0, 1.0) for ll, g in out)
    prefix, target = h._encode_pair(*reqs[0].args)
    # request j of a call draws its masks from the key its own one-by-one call would use (seed + j + 1); the
    # requests share padded forwards, so the values agree up to the rounding of differently tiled GEMMs
    for j in range(2):
        prefix, target = h._encode_pair(*reqs[j].args)
        direct = get_log_likelihood(model, torch.tensor(prefix), torch.tensor(target), mc_num=8, batch_size=4,
                                    mask_id=mask_id, seed=78 + j)
        assert abs(out[j][0] - direct) <= 1e-3 * abs(direct)
    one = h.loglikelihood(reqs[:1])           # a single request takes the one-by-one path: seed 77 + 3
    prefix, target = h._encode_pair(*reqs[0].args)
    assert one[0][0] == get_log_likelihood(model, torch.tensor(prefix), torch.tensor(target), mc_num=8, batch_size=4,
                                           mask_id=mask_id, seed=80)
    texts = h.generate_until([Req("hello", {"until": ["q", "zz"]})])
    assert len(texts) == 1 and isinstance(texts[0], str) and "q" not in texts[0] and len(texts[0]) <= 16


def test_left_padded_row_with_its_own_positions_matches_the_bare_row(lib):
    """What batching requests of different lengths rests on: a row left-padded to a longer batch, its padding
    masked as attention keys and its rotary positions counted from its first real token (positions_from_mask),
    goes through the arithmetic of the bare row -- embedding, norms, projections and the rotated queries and keys of
    the first layer are bit-identical on this model (K = 256: no projection is ever split); attention then sums the
    same terms with other key-tile boundaries, i.e. other bf16 roundings of the probabilities.  Logits: within the
    contract's 2e-2 of the largest logit (measured 1.1e-2 on this random two-layer model), arg-max identical on
    every position whose top-2 margin is clear of that.  The oracle's forward on the bare row agrees with both.
    Without the positions (rotation by column index, as the reference's model does under left padding) the logits
    agree to the same bound."""
    from llada_amd.model import positions_from_mask
    from oracle.model import OracleModel
    model, V = build_tiny()
    z, _ = load("forward_standin.npz")
    d, nl, nh, ff, _ = (int(v) for v in z["cfg"])
    ocfg = OCfg(d_model=d, n_layers=nl, n_heads=nh, mlp_hidden=ff, vocab_size=V)
    W = init_weights(ocfg, seed=int(z["weight_seed"][0]), std=float(z["weight_std"][0]), head_scale=float(z["head_scale"][0]))
    g = np.random.Generator(np.random.PCG64(3))
    lens = [70, 37, 5, 64]                                                # ragged prompts: paddings 0, 33, 65, 6
    L = max(lens)
    ids = np.full((len(lens), L), 7, dtype=np.int64)
    mask = np.zeros((len(lens), L), dtype=np.int64)
    bare = []
    for r, n in enumerate(lens):
        row = g.integers(0, V - 1, size=n, dtype=np.int64)
        ids[r, L - n:], mask[r, L - n:] = row, 1
        bare.append(row)
    ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    pos = positions_from_mask(mask_t)
    assert pos[1].tolist() == [0] * 33 + list(range(37))
    batched = model(ids_t, mask_t, position_ids=pos).logits.float().cpu().numpy()
    by_column = model(ids_t, mask_t).logits.float().cpu().numpy()
    for r, n in enumerate(lens):
        alone = model(torch.from_numpy(bare[r][None, :]).cuda()).logits.float().cpu().numpy()[0]
        got = batched[r, L - n:]
        scale = np.abs(alone).max()
        assert np.abs(got - alone).max() <= 2e-2 * scale, (r, np.abs(got - alone).max() / scale)
        top2 = np.sort(alone, axis=-1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 4e-2 * scale
        assert np.array_equal(got.argmax(-1)[clear], alone.argmax(-1)[clear])
        assert np.abs(by_column[r, L - n:] - alone).max() <= 2e-2 * scale
        ref = OracleModel(W, ocfg, True)(bare[r][None, :])[0]
        assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max()


def test_batched_generate_until_reproduces_the_one_by_one_answers(lib):
    """generate_until with generate_batch_size = 4 (left-padded groups, masked padding, per-row positions) against
    the reference's one-request-at-a-time loop on the same model: the same answers, request for request, in request
    order; and every step of the batched call teacher-forced against the batch-1 forward of the same row."""
    from llada_amd import generate
    from llada_amd.eval_harness import LLaDAEvalHarness
    from llada_amd.model import positions_from_mask
    model, V = build_tiny()
    mask_id = V - 1

    class Tok(CharTokenizer):
        def __call__(self, text):
            return {"input_ids": [ord(c) % (V - 2) + 1 for c in text]}

        def decode(self, ids, skip_special_tokens=False):
            return "".join(chr(97 + int(i) % 26) for i in ids)

    class Req:
        def __init__(self, *args):
            self.args = args

    kw = dict(model=model, tokenizer=Tok(), mask_id=mask_id, batch_size=4, mc_num=8, steps=16, gen_length=16, block_length=8,
              seed=77, pad_id=3)
    questions = ["hello", "a considerably longer question than the others, with many more characters in it", "2+2=",
                 "mid sized prompt here", "x", "another one of medium length!", "the quick brown fox jumps over", ""]
    reqs = [Req(q, {"until": ["q", "zz"]}) for q in questions]
    one_by_one = LLaDAEvalHarness(generate_batch_size=1, **kw).generate_until(reqs)
    batched = LLaDAEvalHarness(generate_batch_size=4, **kw).generate_until(reqs)
    assert len(batched) == len(reqs) and all(isinstance(t, str) for t in batched)
    same = sum(a == b for a, b in zip(one_by_one, batched))
    assert same >= len(reqs) - 1, (one_by_one, batched)        # a near-tied logit of the random model may flip one
    # teacher-forced: the states of a batched call, row by row, against the batch-1 forward on the same tokens
    tok = Tok()
    prompts = [tok(q)["input_ids"] for q in questions[:4]]
    P = max(len(p) for p in prompts)
    ids = torch.full((4, P), 3, dtype=torch.long)
    am = torch.zeros((4, P), dtype=torch.long)
    for r, p in enumerate(prompts):
        ids[r, P - len(p):], am[r, P - len(p):] = torch.tensor(p), 1
    states = []
    generate(model, ids.cuda(), am.cuda(), steps=16, gen_length=16, block_length=8, temperature=0., mask_id=mask_id,
             position_ids="from_mask", step_callback=lambda b, s, x, x0, c: states.append((x.clone(), x0.clone(), c.clone())))
    x_in = torch.cat([ids, torch.full((4, 16), mask_id, dtype=torch.long)], dim=1).cuda()
    am_full = torch.cat([am, torch.ones(4, 16, dtype=torch.long)], dim=1).cuda()
    pos = positions_from_mask(am_full)
    agree = total = 0
    for (x_after, x0, conf) in states:
        lg_b = model(x_in, am_full, position_ids=pos).logits.float()
        for r, p in enumerate(prompts):
            n = len(p) + 16
            alone = model(x_in[r:r + 1, -n:]).logits.float()[0]
            masked = (x_in[r, -n:] == mask_id)
            assert (lg_b[r, -n:] - alone).abs().max() <= 2e-2 * alone.abs().max()
            agree += int((lg_b[r, -n:].argmax(-1) == alone.argmax(-1))[masked].sum())
            total += int(masked.sum())
        x_in = x_after.clone()
    assert agree >= 0.97 * total, (agree, total)      # near-tied logits of the random model flip under 1e-2 noise
