"""Mirror of the evaluation-harness class /root/reference/eval_llada.py:31-279 on the HIP path:
the same constructor keywords, ``loglikelihood`` / ``generate_until`` / ``_encode_pair`` /
``get_loglikelihood`` / ``suffix_greedy_prediction`` with the reference's argument meaning.

The reference class subclasses lm_eval's ``LM`` and is registered as ``llada_dist``; when lm_eval
is importable this class does the same, otherwise it is a plain class with the same methods (the
requests only need an ``.args`` tuple).  Differences, all on the host side:
  * the model is an ``llada_amd.LLaDAModel`` (or any object with the reference's model protocol)
    handed in as ``model=`` or loaded from a local checkpoint directory; nothing is fetched;
  * ranks come from ``torch.distributed`` (one process per GPU), not from accelerate; the
    per-sample barrier of eval_llada.py:277 is kept when a process group exists and skipped
    otherwise (the reference dereferences ``None`` there in a single process);
  * tokenisation does not go through ``datasets.map``.
"""
import ctypes as C

import torch

from . import _lib
from .generate import EOS_ID, EOT_ID, generate
from .get_log_likelihood import get_log_likelihood, get_log_likelihood_many
from .model import LLaDAModel

try:                                                    # eval_llada.py:12-15
    from lm_eval.api.model import LM as _Base
    from lm_eval.api.registry import register_model as _register
except ImportError:                                     # lm_eval is optional
    _Base = object

    def _register(*names):
        return lambda cls: cls


@torch.no_grad()
def suffix_greedy_prediction(model, prefix, target, cfg_scale=0., mask_id=126336, *, trace=None):
    """eval_llada.py:172-194: fill the answer one token per forward, each time committing the
    masked position whose greedy prediction has the highest float32 softmax probability; returns
    a 0-d bool tensor, True iff the filled answer equals ``target``.

    One step = forward (logits of the answer positions only on an LLaDAModel), the fused
    epilogue in float32-confidence mode, and a top-1 transfer; nothing is read back before the
    end.  Ties (torch.sort is not stable in the reference) go to the lowest position."""
    lib = _lib.load()
    device = torch.device(model.device)
    prefix, target = prefix.to(device), target.to(device)
    l1, T = len(prefix), len(target)
    L = l1 + T
    seq = torch.full((1, L), mask_id, dtype=torch.long, device=device)                 # :177
    seq[0, :l1] = prefix                                                               # :180
    fast = isinstance(model, LLaDAModel)
    x0 = torch.empty_like(seq)
    conf = torch.empty((1, L), dtype=torch.float32, device=device)
    sched = torch.ones((1, max(T, 1)), dtype=torch.int32, device=device)
    rows = torch.arange(l1, L, device=device, dtype=torch.int32).contiguous()
    rows2 = torch.cat([rows, rows + L]).contiguous() if cfg_scale > 0. else None
    stream = _lib.stream_ptr()
    ws = torch.empty(lib.llada_rows_workspace_bytes(L), dtype=torch.uint8, device=device)
    for i in range(T):                                                                 # :182
        un_ptr = None
        if cfg_scale > 0.:                                                             # :137-142
            un = seq.clone()
            un[:, :l1] = mask_id
            batch = torch.cat([seq, un])
        else:
            batch = seq
        if fast:
            lg = model.forward_rows(batch, None, rows2 if cfg_scale > 0. else rows)    # :144
            cond = lg[:T]
            if cfg_scale > 0.:
                un_ptr = lg[T:].data_ptr()
            row_pos, nr = rows, T
        else:
            lg = model(batch).logits.to(device=device, dtype=torch.bfloat16).contiguous()
            cond = lg[:1].reshape(L, -1)
            if cfg_scale > 0.:
                un_ptr = lg[1:].reshape(L, -1).data_ptr()
            row_pos, nr = None, L
        args = _lib.RemaskArgs(cond.data_ptr(), un_ptr, float(cfg_scale), _lib.ptr(row_pos), nr, cond.shape[-1],
                               seq.data_ptr(), 1, L, L, mask_id, _lib.REMASK["low_confidence"],
                               _lib.FLAG_CONF_FLOAT32, EOS_ID, EOT_ID, 0.0, None, None, 0, i, 0,
                               x0.data_ptr(), conf.data_ptr(), ws.data_ptr(), ws.numel())
        _lib.check(lib.llada_remask_scores(C.byref(args), stream))                     # :185-188
        _lib.check(lib.llada_topk_transfer(seq.data_ptr(), x0.data_ptr(), conf.data_ptr(), 1, L,
                                           sched.data_ptr(), sched.shape[1], i, stream))   # :189-191
        if trace is not None:
            trace.append(seq[0].clone())
    return torch.all(target == seq[0, l1:])                                            # :192-194


def trim_at_stops(text, stop_seqs):
    """eval_llada.py:268-270."""
    for stop_seq in stop_seqs:
        if stop_seq in text:
            text = text.split(stop_seq)[0]
    return text


@_register("llada_dist")
class LLaDAEvalHarness(_Base):
    def __init__(self, model_path='', mask_id=126336, max_length=4096, batch_size=32, mc_num=128,
                 is_check_greedy=True, cfg=0., steps=1024, gen_length=1024, block_length=1024,
                 remasking='low_confidence', device="cuda", *, model=None, tokenizer=None, process_group=None,
                 seed=1234, rows_per_forward=32, generate_batch_size=1, pad_id=EOS_ID, **kwargs):
        """Keywords as eval_llada.py:33-48.  ``model=`` / ``tokenizer=`` take ready objects; otherwise
        both are loaded from the local directory ``model_path``.  ``generate_batch_size``: requests per
        generate() call in generate_until (1 = one request at a time, as the reference runs them)."""
        if _Base is not object:
            super().__init__()
        self.device = torch.device(device)
        self._rank, self._world_size = 0, 1
        self.process_group = process_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self._rank = torch.distributed.get_rank(process_group)
            self._world_size = torch.distributed.get_world_size(process_group)
        if model is None:
            from .weights import LLaDAWeights
            model = LLaDAModel(LLaDAWeights.from_checkpoint(model_path, self.device))
        self.model = model.eval() if hasattr(model, "eval") else model
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_path, trust_remote_code=True, local_files_only=True)
        self.tokenizer = tokenizer
        self.mask_id = mask_id
        self.mc_num = mc_num
        self.batch_size = int(batch_size)
        assert mc_num % self.batch_size == 0                                           # :95
        self.sampling_eps = 0.
        self.max_length = max_length
        self.is_check_greedy = is_check_greedy
        self.cfg = cfg
        self.steps = steps
        self.gen_length = gen_length
        self.block_length = block_length
        self.remasking = remasking
        self.seed = seed
        self.rows_per_forward = int(rows_per_forward)   # Monte-Carlo rows of several requests per forward
        self.generate_batch_size = max(1, int(generate_batch_size))
        self.pad_id = pad_id
        self._n_ll = 0

    @property
    def rank(self):
        return self._rank

    @property
    def world_size(self):
        return self._world_size

    def _ids(self, text):
        return list(self.tokenizer(text)["input_ids"])

    def get_loglikelihood(self, prefix, target):
        """eval_llada.py:151-170 (the same estimator as get_log_likelihood.py:46-77).  Every call
        draws from its own Philox key so that requests do not share masks."""
        self._n_ll += 1
        return get_log_likelihood(self.model, prefix, target, mc_num=self.mc_num, batch_size=self.batch_size,
                                  cfg_scale=self.cfg, mask_id=self.mask_id, seed=self.seed + self._n_ll)

    def suffix_greedy_prediction(self, prefix, target):
        if not self.is_check_greedy:                                                   # :174-175
            return False
        return suffix_greedy_prediction(self.model, prefix, target, self.cfg, self.mask_id)

    def _encode_pair(self, context, continuation):
        """eval_llada.py:196-208."""
        n_spaces = len(context) - len(context.rstrip())
        if n_spaces > 0:
            continuation = context[-n_spaces:] + continuation
            context = context[:-n_spaces]
        whole_enc = self._ids(context + continuation)
        context_enc = self._ids(context)
        return context_enc, whole_enc[len(context_enc):]

    def loglikelihood(self, requests):
        """eval_llada.py:210-241: [(log-likelihood, 1.0 if the target is the greedy fill else 0.0)]."""
        pairs = [self._encode_pair(req.args[0], req.args[1]) for req in requests]
        assert max((len(p) + len(t) for p, t in pairs), default=0) <= 4096            # :227
        pairs = [(torch.tensor(p, dtype=torch.long), torch.tensor(t, dtype=torch.long)) for p, t in pairs]
        if isinstance(self.model, LLaDAModel) and len(pairs) > 1:
            # the Monte-Carlo rows of different requests share forwards (same masks per request as the
            # one-by-one calls: request j of this call draws from the key its own call would have used)
            seeds = [self.seed + self._n_ll + j + 1 for j in range(len(pairs))]
            self._n_ll += len(pairs)
            lls = get_log_likelihood_many(self.model, pairs, mc_num=self.mc_num, batch_size=self.batch_size,
                                          cfg_scale=self.cfg, mask_id=self.mask_id, seeds=seeds,
                                          max_rows=max(self.batch_size, self.rows_per_forward))
        else:
            lls = [self.get_loglikelihood(prefix, target) for prefix, target in pairs]
        out = []
        for (prefix, target), ll in zip(pairs, lls):
            is_target_greedy_dec = self.suffix_greedy_prediction(prefix, target)
            out.append((ll, 1.0 if is_target_greedy_dec else 0.0))
        return out

    def loglikelihood_rolling(self, requests):
        raise NotImplementedError                                                      # :243-244

    def _finish_answer(self, answer_ids, until):
        """eval_llada.py:267-275: decode, cut at the first stop sequence, strip special tokens."""
        text = self.tokenizer.decode(answer_ids, skip_special_tokens=False)
        text = trim_at_stops(text, until)
        ids = self._ids(text)                                                          # :273-274: strip special tokens
        return self.tokenizer.decode(ids, skip_special_tokens=True)

    def generate_until(self, requests):
        """eval_llada.py:246-279.  The reference runs one request at a time (B = 1, temperature 0), and so does
        this with ``generate_batch_size=1``.  With a larger value (llada_amd.LLaDAModel, cfg 0) requests share
        generate() calls: prompts are left-padded to the longest of their group, the padding is masked as
        attention keys and every row counts its rotary positions from its own first token
        (generate(position_ids="from_mask")), so each row is computed as it is when it runs alone -- same
        arithmetic, up to the rounding of a GEMM tiled for more rows.  Groups are formed in order of prompt
        length (less padding); answers come back in request order."""
        batched = (self.generate_batch_size > 1 and isinstance(self.model, LLaDAModel) and not self.cfg > 0.
                   and len(requests) > 1)
        if not batched:
            out = []
            for req in requests:
                question, until = req.args[0], req.args[1]['until']
                prompt = torch.tensor(self._ids(question), dtype=torch.long)[None, :].to(self.device)
                generated = generate(self.model, prompt, steps=self.steps, gen_length=self.gen_length,
                                     block_length=self.block_length, temperature=0, cfg_scale=self.cfg,
                                     remasking=self.remasking, mask_id=self.mask_id)
                out.append(self._finish_answer(generated[0][prompt.shape[1]:].tolist(), until))
                if self._world_size > 1:
                    torch.distributed.barrier(self.process_group)                      # :277
            return out
        prompts = [self._ids(req.args[0]) for req in requests]
        order = sorted(range(len(requests)), key=lambda i: -len(prompts[i]))
        out = [None] * len(requests)
        for g0 in range(0, len(order), self.generate_batch_size):
            group = order[g0:g0 + self.generate_batch_size]
            P = max(len(prompts[i]) for i in group)
            ids = torch.full((len(group), P), self.pad_id, dtype=torch.long)
            mask = torch.zeros((len(group), P), dtype=torch.long)
            for r, i in enumerate(group):
                n = len(prompts[i])
                if n:                         # an empty prompt is all padding: the answer positions attend only each other
                    ids[r, P - n:] = torch.tensor(prompts[i], dtype=torch.long)
                    mask[r, P - n:] = 1
            generated = generate(self.model, ids.to(self.device), mask.to(self.device), steps=self.steps,
                                 gen_length=self.gen_length, block_length=self.block_length, temperature=0,
                                 cfg_scale=self.cfg, remasking=self.remasking, mask_id=self.mask_id,
                                 position_ids="from_mask")
            rows = generated[:, P:].tolist()
            for r, i in enumerate(group):
                out[i] = self._finish_answer(rows[r], requests[i].args[1]['until'])
            if self._world_size > 1:
                torch.distributed.barrier(self.process_group)
        return out
