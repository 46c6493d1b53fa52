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
