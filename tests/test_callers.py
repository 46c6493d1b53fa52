"""The callers restated in llada_amd.callers (chat.py:7-40; dllm.py:371-405, 503-549) and the local
checkpoint loader.  Host logic runs on the CPU; the end-to-end cases need the GPU."""
import json

import numpy as np
import pytest
import torch

MASK_ID, EOS_ID = 126336, 126081


class ToyTokenizer:
    """Character tokenizer over the LLaDA id space: id = 1000 + code point; BOS = 1; pad = EOS."""
    padding_side = "right"
    pad_token_id = EOS_ID

    def _enc(self, text):
        return [1] + [1000 + ord(c) for c in text]

    def __call__(self, text, add_special_tokens=True):
        ids = self._enc(text)
        return {"input_ids": ids if add_special_tokens else ids[1:]}

    def encode(self, text):
        return self._enc(text)

    def apply_chat_template(self, messages, add_generation_prompt=True, tokenize=False):
        s = "".join(f"<{m['role']}>{m['content']}</>" for m in messages)
        return s + ("<assistant>" if add_generation_prompt else "")

    def batch_encode_plus(self, texts, padding=True, return_tensors="pt"):
        enc = [self._enc(t) for t in texts]
        n = max(len(e) for e in enc)
        rows = [[self.pad_token_id] * (n - len(e)) + e if self.padding_side == "left" else e + [self.pad_token_id] * (n - len(e))
                for e in enc]
        return {"input_ids": torch.tensor(rows, dtype=torch.long)}

    def decode(self, ids, skip_special_tokens=False):
        out = []
        for i in (int(v) for v in ids):
            if 1000 <= i < 1000 + 0x110000 and i < 100000:
                out.append(chr(97 + (i - 1000) % 26) if i - 1000 > 0x7f else chr(i - 1000))
            elif not skip_special_tokens:
                out.append("<%d>" % i)
            elif i < 1000 or i >= 126000:
                continue
            else:
                out.append(chr(97 + i % 26))
        return "".join(out)

    def batch_decode(self, rows, skip_special_tokens=False):
        return [self.decode(r, skip_special_tokens) for r in rows]


def test_checkpoint_loader_roundtrip(tmp_path):
    """A checkpoint directory written with the LLaDA tensor names loads into the same packed
    weights as the in-memory constructor."""
    from safetensors.torch import save_file

    from llada_amd import LLaDAConfig, LLaDAWeights
    from oracle.model import LLaDAConfig as OCfg
    from oracle.model import init_weights
    ocfg = OCfg(d_model=256, n_layers=2, n_heads=2, mlp_hidden=192, vocab_size=512)
    W = init_weights(ocfg, seed=9, std=0.05, head_scale=2.0)
    tensors = {}
    for k, v in W.items():
        tensors[f"model.transformer.{k}.weight"] = torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16)
    half = len(tensors) // 2
    names = sorted(tensors)
    save_file({n: tensors[n] for n in names[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({n: tensors[n] for n in names[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps({"d_model": 256, "n_layers": 2, "n_heads": 2, "mlp_hidden_size": 192,
                                                       "embedding_size": 512, "vocab_size": 500, "rope_theta": 500000.0,
                                                       "rms_norm_eps": 1e-5, "mask_token_id": 511}))
    a = LLaDAWeights.from_checkpoint(str(tmp_path), "cpu")
    cfg = LLaDAConfig(d_model=256, n_layers=2, n_heads=2, mlp_hidden=192, vocab_size=512, mask_token_id=511)
    b = LLaDAWeights.from_numpy(cfg, W, "cpu")
    assert a.cfg.vocab_size == 512 and a.cfg.mask_token_id == 511 and a.cfg.mlp_hidden == 192
    assert torch.equal(a.wte, b.wte) and torch.equal(a.lm_head, b.lm_head) and torch.equal(a.ln_f, b.ln_f)
    for i in range(2):
        for name in ("attn_norm", "wqkv", "wo", "ff_norm", "wgu", "wdown"):
            assert torch.equal(getattr(a, name)[i], getattr(b, name)[i]), (name, i)


def test_message_converters():
    """OpenCompass hands the wrappers PromptLists ({"role": HUMAN | BOT | SYSTEM, "prompt": text}); the expected
    outputs below are what dllm.py:461-501 produces for these inputs (worked out by hand from its rules: role map,
    empty prompts dropped, consecutive turns of one role joined with a newline; base models: plain concatenation)."""
    from llada_amd.callers import chat_messages, flat_prompt
    assert chat_messages("plain question") == [{"role": "user", "content": "plain question"}]
    pl = [{"role": "SYSTEM", "prompt": "be brief"}, {"role": "HUMAN", "prompt": "Q1"}, {"role": "HUMAN", "prompt": "Q1b"},
          {"role": "BOT", "prompt": ""}, {"role": "BOT", "prompt": "A1"}, {"role": "HUMAN", "prompt": "Q2"}]
    assert chat_messages(pl) == [{"role": "system", "content": "be brief"}, {"role": "user", "content": "Q1\nQ1b"},
                                 {"role": "assistant", "content": "A1"}, {"role": "user", "content": "Q2"}]
    assert chat_messages(pl, merge_role=False)[1:3] == [{"role": "user", "content": "Q1"}, {"role": "user", "content": "Q1b"}]
    assert chat_messages([{"role": "BOT", "prompt": ""}], skip_empty_prompt=False) == [{"role": "assistant", "content": ""}]
    # already-converted chat messages pass through
    done = [{"role": "user", "content": "x"}, {"role": "assistant", "content": "y"}]
    assert chat_messages(done) == done
    with pytest.raises(KeyError):
        chat_messages([{"role": "NARRATOR", "prompt": "x"}])          # the reference's role table raises KeyError too
    with pytest.raises(TypeError):
        chat_messages([{"role": "HUMAN"}])
    with pytest.raises(TypeError):
        chat_messages(["not a dict"])
    assert flat_prompt("as is") == "as is"
    assert flat_prompt(pl) == "be briefQ1Q1bA1Q2"
    with pytest.raises(TypeError):
        flat_prompt([{"role": "HUMAN"}])


def test_wrappers_convert_prompt_lists_before_tokenizing():
    """The wrappers' prompt assembly on PromptList requests, no model needed (dllm.py:373-374, 509-510, 551-554)."""
    from llada_amd.callers import LLaDABaseModel, LLaDAModel
    tok = ToyTokenizer()
    pl = [{"role": "HUMAN", "prompt": "Q"}, {"role": "BOT", "prompt": "A"}, {"role": "HUMAN", "prompt": "R"}]
    chat = LLaDAModel(model=object(), tokenizer=tok)
    assert chat._prompts(["hi", pl]) == ["<user>hi</><assistant>", "<user>Q</><assistant>A</><user>R</><assistant>"]
    base = LLaDABaseModel(model=object(), tokenizer=tok)
    assert base._prompts(["hi", pl]) == ["hi", "QAR"]
    assert base.get_token_len(pl) == 4 and base.get_token_len(pl, add_special_tokens=False) == 3


@pytest.fixture(scope="module")
def small_full_vocab_model():
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    cfg = LLaDAConfig(d_model=256, n_layers=2, n_heads=2, mlp_hidden=512)
    return LLaDAModel(LLaDAWeights.random(cfg, "cuda", seed=3))


@pytest.mark.gpu
def test_chat_turns(lib, small_full_vocab_model):
    from llada_amd import generate
    from llada_amd.callers import chat_turn
    model, tok = small_full_vocab_model, ToyTokenizer()
    ans1, ctx1 = chat_turn(model, tok, None, "hi there", gen_length=32, steps=32, block_length=8)
    first = torch.tensor(tok(tok.apply_chat_template([{"role": "user", "content": "hi there"}]))["input_ids"])
    assert isinstance(ans1, str) and ctx1.dim() == 2 and ctx1.shape[0] == 1
    assert torch.equal(ctx1[0, :len(first)].cpu(), first) and not (ctx1 == EOS_ID).any() and not (ctx1 == MASK_ID).any()
    # the turn is exactly generate() on the assembled prompt (chat.py:32)
    ref = generate(model, first[None, :].cuda(), steps=32, gen_length=32, block_length=8, temperature=0., cfg_scale=0.,
                   remasking='low_confidence')
    assert torch.equal(ctx1, ref[ref != EOS_ID].unsqueeze(0))
    ans2, ctx2 = chat_turn(model, tok, ctx1, "and then?", gen_length=32, steps=32, block_length=8)
    second = torch.tensor(tok(tok.apply_chat_template([{"role": "user", "content": "and then?"}]))["input_ids"])
    n1 = ctx1.shape[1]
    assert torch.equal(ctx2[0, :n1], ctx1[0])                                   # the context grows (chat.py:30)
    assert torch.equal(ctx2[0, n1:n1 + len(second) - 1].cpu(), second[1:])      # BOS of later turns is dropped


@pytest.mark.gpu
def test_opencompass_wrappers(lib, small_full_vocab_model):
    from llada_amd import generate
    from llada_amd.callers import LLaDABaseModel, LLaDAModel
    model, tok = small_full_vocab_model, ToyTokenizer()
    kw = dict(model=model, tokenizer=tok, gen_steps=16, gen_length=16, gen_blocksize=8, batch_size_=3,
              diff_confidence_eos_eot_inf=True, diff_logits_eos_inf=True)
    inputs = ["short", "a much longer question than the first", "xy"]
    # OpenCompass PromptLists (dllm.py:461-501) next to plain strings and ready chat messages
    prompt_list = [{"role": "HUMAN", "prompt": "x"}, {"role": "BOT", "prompt": "y"}, {"role": "HUMAN", "prompt": "z"}]
    chat_inputs = inputs[:1] + [prompt_list] + [[{"role": "user", "content": "x"}, {"role": "assistant", "content": "y"},
                                                 {"role": "user", "content": "z"}]]
    chat_out = LLaDAModel(**kw).generate(chat_inputs, max_out_len=16)
    assert len(chat_out) == 3 and all(isinstance(r, str) for r in chat_out)
    base = LLaDABaseModel(stop_words=["q", "zz"], **kw)
    as_lists = base.generate([[{"role": "HUMAN", "prompt": "sho"}, {"role": "HUMAN", "prompt": "rt"}], inputs[1],
                              [{"role": "HUMAN", "prompt": "xy"}]], max_out_len=16)
    out = base.generate(inputs, max_out_len=16)
    assert as_lists == out                                                      # "sho" + "rt" is the prompt "short"
    assert len(out) == 3 and all("q" not in r for r in out)
    # same tokens as generate() on the left-padded batch without an attention mask (dllm.py:516-530)
    tok.padding_side = "left"
    prompt = tok.batch_encode_plus(["short", "a much longer question than the first", "xy"], padding=True)["input_ids"]
    assert (prompt[0, :5] == EOS_ID).all()
    x = generate(model, prompt.cuda(), steps=16, gen_length=16, block_length=8, temperature=0., cfg_scale=0.,
                 remasking='low_confidence', confidence_eos_eot_inf=True, logits_eos_inf=True)
    full = [tok.decode(x[i, -16:], skip_special_tokens=True) for i in range(3)]
    for got, whole in zip(out, full):
        assert whole.startswith(got)
    assert base.get_token_len("abc") == 4 and base.get_token_len("abc", add_special_tokens=False) == 3
    with pytest.raises(NotImplementedError):
        base.get_ppl(["x"])


class ProtocolModel:
    """A foreign model that only honours the reference's protocol (generate.py:60,88,92); its logits are a
    closed-form function of the call number, regenerated on the oracle's side."""

    def __init__(self, seed, device):
        self.device, self.seed, self.calls = torch.device(device), seed, 0

    def logits_np(self, shape):
        from tests.golden_util import fake_model_logits
        lg = fake_model_logits(self.seed, self.calls, shape + (VOCAB,))
        self.calls += 1
        return lg

    def __call__(self, x, attention_mask=None):
        class Out:
            pass
        o = Out()
        o.logits = torch.from_numpy(self.logits_np(tuple(x.shape))).to(self.device).to(torch.bfloat16)
        return o


VOCAB = 126464


@pytest.mark.gpu
def test_wrappers_against_the_oracle_sampler(lib):
    """The OpenCompass-style wrapper and chat_turn on a foreign protocol model against the ORACLE's sampler
    (oracle/sampler.py::generate, pinned to the reference's recorded runs) fed the same logits: an independent
    check of the prompt assembly, the left padding without attention mask, both EOS flags and the decode of
    the last gen_length ids."""
    from oracle import sampler as osamp
    from oracle.bf16 import round_bf16
    from llada_amd.callers import LLaDABaseModel, chat_turn
    tok = ToyTokenizer()

    class NpModel:
        def __init__(self, seed):
            self.inner = ProtocolModel(seed, "cpu")

        def __call__(self, x, attention_mask=None):
            return round_bf16(self.inner.logits_np(tuple(x.shape)))

    inputs = ["short", "a much longer question than the first", "xy"]
    base = LLaDABaseModel(model=ProtocolModel(41, "cuda"), tokenizer=tok, gen_steps=8, gen_length=16, gen_blocksize=8,
                          batch_size_=3, diff_confidence_eos_eot_inf=True, diff_logits_eos_inf=True)
    got = base.generate(inputs, max_out_len=16)
    tok.padding_side = "left"
    prompt = tok.batch_encode_plus(inputs, padding=True)["input_ids"].numpy()
    want = osamp.generate(NpModel(41), prompt, None, steps=8, gen_length=16, block_length=8, temperature=0., cfg_scale=0.,
                          remasking="low_confidence", mask_id=MASK_ID, logits_eos_inf=True, confidence_eos_eot_inf=True)
    assert got == [tok.decode(want[i, -16:], skip_special_tokens=True) for i in range(3)]

    ans, ctx = chat_turn(ProtocolModel(43, "cuda"), tok, None, "hi there", gen_length=16, steps=16, block_length=8)
    first = np.array([tok(tok.apply_chat_template([{"role": "user", "content": "hi there"}]))["input_ids"]])
    want = osamp.generate(NpModel(43), first, None, steps=16, gen_length=16, block_length=8, temperature=0., cfg_scale=0.,
                          remasking="low_confidence", mask_id=MASK_ID)
    assert ans == tok.decode(want[0, first.shape[1]:], skip_special_tokens=True)
    assert np.array_equal(ctx.cpu().numpy()[0], want[0][want[0] != EOS_ID])
