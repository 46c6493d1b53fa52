"""The oracle's restatement of the evaluation-harness helpers (eval_llada.py:172-208, 267-275) and the
host-side logic of llada_amd.eval_harness that needs no GPU.  Parity unpinned: eval_llada.py cannot
be imported offline (lm_eval is absent), so these check the restatement against the semantics the
source text states (one committed token per forward, monotone fill, whitespace rule)."""
import numpy as np
import pytest

from oracle import harness as oh
from tests.golden_util import FakeModel

MASK = 1023


class CharTokenizer:
    """Toy tokenizer: one id per character (its code point), 0 is a special token."""

    def __call__(self, text):
        return {"input_ids": [ord(c) for c in text]}

    def decode(self, ids, skip_special_tokens=False):
        return "".join("<s>" if (i == 0 and not skip_special_tokens) else ("" if i == 0 else chr(int(i))) for i in ids)


def test_suffix_greedy_commits_one_token_per_forward():
    model = FakeModel(seed=5, vocab=1024)
    prefix = np.arange(7, dtype=np.int64)
    target = np.arange(100, 111, dtype=np.int64)
    trace = []
    ok, seq = oh.suffix_greedy_prediction(lambda x, am: model(x), prefix, target, 0.0, MASK, trace=trace)
    assert model.calls == len(target) and len(trace) == len(target)
    prev = np.concatenate([prefix, np.full(len(target), MASK)])
    for n, cur in enumerate(trace):
        assert (cur == MASK).sum() == len(target) - n - 1          # exactly one more position each step
        keep = prev != MASK
        assert np.array_equal(cur[keep], prev[keep])               # committed tokens never change
        prev = cur
    assert ok == bool(np.array_equal(seq[7:], target))


def test_suffix_greedy_true_when_model_predicts_target():
    target = np.array([5, 9, 200, 3], dtype=np.int64)
    prefix = np.array([1, 2], dtype=np.int64)

    def model(x, am):
        lg = np.zeros(x.shape + (256,), dtype=np.float32)
        for t, tok in enumerate(target):
            lg[:, 2 + t, tok] = 4.0 + t
        return lg
    ok, seq = oh.suffix_greedy_prediction(model, prefix, target, 0.0, 255)
    assert ok and np.array_equal(seq, np.concatenate([prefix, target]))
    ok2, _ = oh.suffix_greedy_prediction(model, prefix, target[::-1].copy(), 0.0, 255)
    assert not ok2


def test_encode_pair_moves_trailing_spaces():
    tok = lambda s: [ord(c) for c in s]                                    # noqa: E731
    ctx, cont = oh.encode_pair(tok, "Q: why?  ", "because")
    assert ctx == tok("Q: why?") and cont == tok("  because")
    ctx, cont = oh.encode_pair(tok, "abc", " d")
    assert ctx == tok("abc") and cont == tok(" d")


def test_trim_at_stops():
    assert oh.trim_at_stops("answer 42\n\nQuestion: next", ["\n\n", "Question:"]) == "answer 42"
    assert oh.trim_at_stops("no stop here", ["###"]) == "no stop here"


def test_harness_host_logic_matches_oracle():
    from llada_amd.eval_harness import LLaDAEvalHarness, trim_at_stops
    h = LLaDAEvalHarness(model=object(), tokenizer=CharTokenizer(), device="cpu", batch_size=4, mc_num=8,
                         is_check_greedy=False)
    tok = lambda s: [ord(c) for c in s]                                    # noqa: E731
    for ctx, cont in (("Q: why?  ", "because"), ("abc", " d"), ("x \n", "y")):
        assert h._encode_pair(ctx, cont) == oh.encode_pair(tok, ctx, cont)
    assert h.suffix_greedy_prediction(None, None) is False                  # eval_llada.py:174-175
    assert h.rank == 0 and h.world_size == 1
    with pytest.raises(NotImplementedError):
        h.loglikelihood_rolling([])
    with pytest.raises(AssertionError):                                     # eval_llada.py:95
        LLaDAEvalHarness(model=object(), tokenizer=CharTokenizer(), device="cpu", batch_size=3, mc_num=8)
    for text, stops in (("a b\n\nc", ["\n\n"]), ("abc", ["zz"]), ("aXbYc", ["Y", "X"])):
        assert trim_at_stops(text, stops) == oh.trim_at_stops(text, stops)
