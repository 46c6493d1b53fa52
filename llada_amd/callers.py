"""The thin callers that sit directly on generate() in the reference, restated on the HIP path so
they can be switched over by changing an import:

  * ``chat`` / ``chat_turn``            -- /root/reference/chat.py:7-40 (multi-turn prompt assembly,
                                           generation settings, EOS stripping of the running context);
  * ``LLaDAModel`` / ``LLaDABaseModel`` -- the ``generate`` methods of the OpenCompass wrappers,
                                           /root/reference/opencompass/opencompass/models/dllm.py:371-405
                                           and :503-549 (left-padded batch encode, no attention mask,
                                           decode of the last ``gen_length`` tokens, stop words).
                                           Requests arrive as OpenCompass hands them over: a string, or a
                                           PromptList of {"role": HUMAN | BOT | SYSTEM, "prompt": text} items,
                                           turned into chat messages / one flat string as dllm.py:461-501 does
                                           (``chat_messages`` / ``flat_prompt`` below); lists that are already
                                           {"role", "content"} chat messages are accepted as well.

Only the parts that touch the sampler are restated; the OpenCompass base class (mmengine registry,
template parsers, perplexity helpers that the reference itself disables at dllm.py:423) is not.
Model and tokenizer are ready objects or come from a local directory; nothing is downloaded.
"""
import torch

from .generate import EOS_ID, generate


def load_local(path, device="cuda"):
    """(model, tokenizer) from a local checkpoint directory."""
    from transformers import AutoTokenizer

    from .model import LLaDAModel as HipModel
    from .weights import LLaDAWeights
    model = HipModel(LLaDAWeights.from_checkpoint(path, device))
    tokenizer = AutoTokenizer.from_pretrained(path, trust_remote_code=True, local_files_only=True)
    return model, tokenizer


def chat_turn(model, tokenizer, prompt, user_input, gen_length=128, steps=128, block_length=32):
    """One iteration of the loop at chat.py:19-40.  ``prompt`` is the running context (None on the
    first turn).  Returns (answer text, new running context with every EOS removed)."""
    device = torch.device(model.device)
    m = [{"role": "user", "content": user_input}]
    text = tokenizer.apply_chat_template(m, add_generation_prompt=True, tokenize=False)            # :23
    input_ids = torch.tensor(tokenizer(text)['input_ids']).to(device).unsqueeze(0)                  # :24-25
    prompt = input_ids if prompt is None else torch.cat([prompt, input_ids[:, 1:]], dim=1)          # :27-30
    out = generate(model, prompt, steps=steps, gen_length=gen_length, block_length=block_length, temperature=0.,
                   cfg_scale=0., remasking='low_confidence')                                        # :32
    answer = tokenizer.batch_decode(out[:, prompt.shape[1]:], skip_special_tokens=True)[0]          # :34
    return answer, out[out != EOS_ID].unsqueeze(0)                                                  # :38


def chat(model_path, device="cuda", input_fn=input, print_fn=print):
    """chat.py:7-40 against a local checkpoint directory."""
    model, tokenizer = load_local(model_path, device)
    gen_length, steps = 128, 128
    print_fn('*' * 66)
    print_fn(f'**  Answer Length: {gen_length}  |  Sampling Steps: {steps}  **')
    print_fn('*' * 66)
    prompt = None
    while True:
        user_input = input_fn("Enter your question: ")
        answer, prompt = chat_turn(model, tokenizer, prompt, user_input, gen_length, steps)
        print_fn(f"Bot's reply: {answer}")
        print_fn('-----------------------------------------------------------------------')


_OPENCOMPASS_ROLES = {"HUMAN": "user", "BOT": "assistant", "SYSTEM": "system"}


def chat_messages(request, merge_role=True, skip_empty_prompt=True):
    """One request -> the message list tokenizer.apply_chat_template takes (what dllm.py:461-488 builds per
    input): a string is a single user turn; a PromptList item {"role": "HUMAN" | "BOT" | "SYSTEM", "prompt": text}
    becomes {"role": "user" | "assistant" | "system", "content": text}; items with an empty prompt are dropped;
    consecutive turns of one role are joined with a newline.  Items already in {"role", "content"} form pass
    through (an extension).  Anything else raises with the offending item named."""
    if isinstance(request, str):
        return [{"role": "user", "content": request}]
    turns = []
    for item in request:
        if not isinstance(item, dict) or "role" not in item or not ("prompt" in item or "content" in item):
            raise TypeError(f"chat request item {item!r}: expected a dict with 'role' and 'prompt' (or 'content')")
        if "prompt" in item:
            if item["role"] not in _OPENCOMPASS_ROLES:
                raise KeyError(item["role"])                     # the reference's role table raises the same
            role, text = _OPENCOMPASS_ROLES[item["role"]], item["prompt"]
            if skip_empty_prompt and not text:
                continue
        else:
            role, text = item["role"], item["content"]
        if merge_role and turns and turns[-1]["role"] == role:
            turns[-1]["content"] += "\n" + text
        else:
            turns.append({"role": role, "content": text})
    return turns


def flat_prompt(request):
    """One request -> the plain string a base model is prompted with (dllm.py:491-501): a string as it is, a
    PromptList's texts concatenated without separators."""
    if isinstance(request, str):
        return request
    parts = []
    for item in request:
        if not isinstance(item, dict) or not ("prompt" in item or "content" in item):
            raise TypeError(f"base-model request item {item!r}: expected a dict with 'prompt'")
        parts.append(item["prompt"] if "prompt" in item else item["content"])
    return "".join(parts)


class LLaDAModel:
    """The sampler-facing part of dllm.py:69-405 (chat-template model)."""

    def __init__(self, path='', model=None, tokenizer=None, max_seq_len=2048, stop_words=(), cfg=0, temperature=0.,
                 remasking='low_confidence', mask_id=126336, padding_id=126081, mc_num=1, gen_steps=512,
                 gen_length=512, gen_blocksize=512, batch_size_=1, diff_confidence_eos_eot_inf=False,
                 diff_logits_eos_inf=False, device="cuda", verbose=False, **kwargs):
        if model is None or tokenizer is None:
            model, tokenizer = load_local(path, device)
        self.model, self.tokenizer = model, tokenizer
        self.max_seq_len = max_seq_len
        self.stop_words = list(stop_words)
        self.cfg, self.mc_num = cfg, mc_num                                                # dllm.py:171-184
        self.batch_size = batch_size_
        self.gen_steps, self.gen_length, self.gen_blocksize = gen_steps, gen_length, gen_blocksize
        self.temperature, self.remasking = temperature, remasking
        self.padding_id, self.mask_id = padding_id, mask_id
        self.diff_confidence_eos_eot_inf = diff_confidence_eos_eot_inf
        self.diff_logits_eos_inf = diff_logits_eos_inf
        self.verbose = verbose

    def get_token_len(self, prompt):
        return len(self.tokenizer.encode(prompt))                                          # dllm.py:360-369

    def _prompts(self, inputs):
        """One chat-template prompt per request (dllm.py:373-374)."""
        return [self.tokenizer.apply_chat_template(chat_messages(m), add_generation_prompt=True, tokenize=False)
                for m in inputs]

    def _sample(self, prompt_texts):
        self.tokenizer.padding_side = "left"                                               # dllm.py:380
        prompt = self.tokenizer.batch_encode_plus(prompt_texts, padding=True, return_tensors='pt')['input_ids']
        x = generate(model=self.model, prompt=prompt.to(self.model.device), steps=self.gen_steps,
                     gen_length=self.gen_length, block_length=self.gen_blocksize, temperature=self.temperature,
                     cfg_scale=self.cfg, remasking=self.remasking, mask_id=self.mask_id,
                     confidence_eos_eot_inf=self.diff_confidence_eos_eot_inf,
                     logits_eos_inf=self.diff_logits_eos_inf)                              # dllm.py:382-394 (no attention_mask)
        responses = [self.tokenizer.decode(x[i, -self.gen_length:], skip_special_tokens=True)
                     for i in range(prompt.shape[0])]                                      # dllm.py:398-399
        if self.verbose:
            for i, r in enumerate(responses):
                print(f'Response {i}:', r)
        return responses

    def generate(self, inputs, max_out_len=None):
        """dllm.py:371-405."""
        return self._sample(self._prompts(inputs))

    def get_ppl(self, inputs, mask_length=None):
        raise NotImplementedError('Please use `lmeval` toolkit instead.')                  # dllm.py:423


class LLaDABaseModel(LLaDAModel):
    """dllm.py:503-554 (base model: plain concatenated prompt, stop-word trimming)."""

    def _prompts(self, inputs):
        return [flat_prompt(m) for m in inputs]                                             # dllm.py:509-510

    def generate(self, inputs, max_out_len=None):
        responses = self._sample(self._prompts(inputs))
        stopping_criteria = set(self.stop_words)                                           # dllm.py:541-548
        for i, response in enumerate(responses):
            for stop_word in stopping_criteria:
                if stop_word in response:
                    response = response.split(stop_word)[0]
                    break
            responses[i] = response
        return responses

    def get_token_len(self, prompt, add_special_tokens=True):
        return len(self.tokenizer(flat_prompt(prompt), add_special_tokens=add_special_tokens)['input_ids'])   # dllm.py:551-554
