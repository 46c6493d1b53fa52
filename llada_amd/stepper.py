"""Step-by-step driver of the sampler for interactive callers: the loop body of generate()
(/root/reference/generate.py:80-134) split into the pieces an interactive front end asks for one at a time,
on the same kernels.  It offers what /root/reference/viz/generation_engine.py drives its own private copy of
the loop with: a forward pass that leaves every position's candidate token and confidence
(`step_generation`, viz/generation_engine.py:160-236), committing a caller-chosen number of tokens from the
cached scores without another forward (`select_tokens_only`, :238-311), caller-forced tokens
(`manual_selections`, :174-179, :256-261) and going back to an earlier state (`rewind_to_step`, :336-341).
The arithmetic is generate.py's (bf16 softmax confidence, the five remasking rules, float64 Gumbel noise),
not the visualiser's float32 variant; driven with the reference's schedule it reproduces generate() exactly.
Batch 1 like the visualiser, or any batch."""
import ctypes as C

import torch

from . import _lib
from .generate import EOS_ID, EOT_ID
from .model import LLaDAModel, check_token_ids


class GenerationStepper:
    def __init__(self, model, prompt, attention_mask=None, gen_length=128, block_length=32, temperature=0.,
                 cfg_scale=0., remasking='low_confidence', mask_id=126336, logits_eos_inf=False,
                 confidence_eos_eot_inf=False, seed=0, max_history=None):
        assert gen_length % block_length == 0                                             # generate.py:68
        if remasking not in _lib.REMASK:
            raise NotImplementedError(remasking)                                          # :122-123
        if cfg_scale > 0.0 and attention_mask is None:
            raise UnboundLocalError("cannot access local variable 'attention_mask_' where it is not associated with a value")
        self.lib = _lib.load()
        self.model, self.device = model, torch.device(model.device)
        self.fast = isinstance(model, LLaDAModel)
        if self.fast:
            check_token_ids(prompt, model.cfg.vocab_size, mask_id, "prompt")
        B, P = prompt.shape
        self.B, self.P, self.L = B, P, P + gen_length
        self.block_length, self.num_blocks = block_length, gen_length // block_length
        self.temperature, self.cfg_scale, self.remasking = float(temperature), float(cfg_scale), remasking
        self.mask_id, self.seed = mask_id, seed
        self.flags = (_lib.FLAG_LOGITS_EOS_INF if logits_eos_inf else 0) | (_lib.FLAG_CONF_EOS_EOT_INF if confidence_eos_eot_inf else 0)
        self.x = torch.full((B, self.L), mask_id, dtype=torch.long, device=self.device)   # :60-61
        self.x[:, :P] = prompt.to(self.device)
        self.attention_mask = None
        if attention_mask is not None:                                                    # :63-64
            self.attention_mask = torch.cat([attention_mask.to(self.device),
                                             torch.ones((B, gen_length), dtype=attention_mask.dtype, device=self.device)], dim=-1)
        self.prompt_index = self.x != mask_id                                             # :66
        self.x0 = torch.empty_like(self.x)
        self.conf = torch.empty((B, self.L), dtype=torch.float32, device=self.device)
        self.ws = torch.empty(self.lib.llada_rows_workspace_bytes(B * self.L), dtype=torch.uint8, device=self.device)
        self.step_index = 0            # forwards run so far (the Philox step counter)
        self.scores_valid = False      # x0 / conf belong to the current x
        # state after every select(): (tokens, Philox step counter), so that a replay after rewind() draws the
        # noise the original run drew.  ``max_history``: keep only that many most recent states (None = all;
        # each is a [B, L] int64 clone); state 0, the fully masked answer, is always kept.
        self.max_history = max_history
        self.history = {0: (self.x.clone(), 0)}
        self._n_states = 1
        self._block = None             # cached index of the current block; None = recompute (one host read)

    # -- state ------------------------------------------------------------------------------------------
    @property
    def block(self):
        """Index of the first block that still holds a mask token (num_blocks when none does).  Computed once per
        change of the tokens (one host read), then cached."""
        if self._block is None:
            gen = self.x[:, self.P:].reshape(self.B, self.num_blocks, self.block_length)
            open_blocks = (gen == self.mask_id).any(dim=2).any(dim=0)
            idx = torch.nonzero(open_blocks)
            self._block = int(idx[0].item()) if idx.numel() else self.num_blocks
        return self._block

    def _force(self, manual_selections, drop_scores):
        """Writes the caller's own tokens ({column: id} for row 0, or {(row, column): id}) into x.  Ids go through
        the same vocabulary check as the prompt (the embedding kernel would clamp an out-of-range id silently)."""
        if not manual_selections:
            return
        items = []
        for key, tok in manual_selections.items():
            r, c = key if isinstance(key, tuple) else (0, key)
            if 0 <= c < self.L:
                if not 0 <= r < self.B:
                    raise IndexError(f"manual selection row {r} is outside the batch of {self.B}")
                items.append((r, c, int(tok)))
        if not items:
            return
        if self.fast:
            check_token_ids(torch.tensor([t for _, _, t in items]), self.model.cfg.vocab_size, None, "manual_selections")
        for r, c, tok in items:
            self.x[r, c] = tok
            if drop_scores:
                self.conf[r, c] = float("-inf")            # no longer a candidate
        self._block = None

    @property
    def done(self):
        return self.block >= self.num_blocks

    def masked_in_block(self):
        """[B] number of mask tokens left in the current block."""
        b = self.block
        if b >= self.num_blocks:
            return torch.zeros(self.B, dtype=torch.long, device=self.device)
        lo = self.P + b * self.block_length
        return (self.x[:, lo:lo + self.block_length] == self.mask_id).sum(dim=1)

    # -- generate.py:81-128 -------------------------------------------------------------------------------
    def forward(self):
        """One forward pass + the fused epilogue on the current tokens: afterwards ``x0`` holds every masked
        position's candidate (generate.py:127) and ``confidence`` its score (:128; -inf outside the current
        block and on revealed positions).  Returns (x0, confidence), live device tensors."""
        b = self.block
        if b >= self.num_blocks:
            raise StopIteration("generation is complete")
        bs, be = self.P + b * self.block_length, self.P + (b + 1) * self.block_length
        B, L = self.B, self.L
        x = self.x
        if self.cfg_scale > 0.0:                                                          # :82-90
            x_in = torch.cat([x, torch.where(self.prompt_index, self.mask_id, x)], dim=0)
            am = torch.cat([self.attention_mask, self.attention_mask], dim=0)
        else:
            x_in, am = x, self.attention_mask
        un_ptr, row_pos = None, None
        if self.fast:
            # logits for every position left of the block's end that can still be committed (see generate())
            row_lo = bs if not bool((x[:, :bs] == self.mask_id).any().item()) else 0
            rows = (torch.arange(B, device=self.device, dtype=torch.int32)[:, None] * L
                    + torch.arange(row_lo, be, device=self.device, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
            n_rows = rows.numel()
            rows_in = torch.cat([rows, rows + B * L]).contiguous() if self.cfg_scale > 0.0 else rows
            lg = self.model.forward_rows(x_in, am, rows_in)
            cond, row_pos, nr = lg[:n_rows], rows, n_rows
            if self.cfg_scale > 0.0:
                un_ptr = lg[n_rows:].data_ptr()
        else:
            lg = self.model(x_in, attention_mask=am).logits.to(device=self.device, dtype=torch.bfloat16).contiguous()
            cond, nr = lg[:B].reshape(B * L, -1), B * L
            if self.cfg_scale > 0.0:
                un_ptr = lg[B:].reshape(B * L, -1).data_ptr()
        self._keep = lg                                                                   # alive until the kernel has run
        args = _lib.RemaskArgs(cond.data_ptr(), un_ptr, self.cfg_scale, _lib.ptr(row_pos), nr, cond.shape[-1], x.data_ptr(),
                               B, L, be, self.mask_id, _lib.REMASK[self.remasking], self.flags, EOS_ID, EOT_ID,
                               self.temperature, None, None, self.seed, self.step_index, 0, self.x0.data_ptr(),
                               self.conf.data_ptr(), self.ws.data_ptr(), self.ws.numel())
        _lib.check(self.lib.llada_remask_scores(C.byref(args), _lib.stream_ptr()))        # :94-128
        self.step_index += 1
        self.scores_valid = True
        return self.x0, self.conf

    # -- generate.py:130-134 with a caller-chosen k -------------------------------------------------------
    def select(self, tokens_to_select=1, manual_selections=None):
        """Commit tokens from the cached scores, without a forward pass: first the caller's own choices
        (``manual_selections``: {column: token id} for row 0, or {(row, column): token id}), then the
        ``tokens_to_select`` most confident candidates of the current block per row (an int or one int per
        row; never more than the row has mask tokens left in the block).  Returns the tokens [B, L]."""
        if not self.scores_valid:
            raise ValueError("no cached scores: run forward() first")
        self._force(manual_selections, True)
        left = (self.x == self.mask_id) & ~torch.isneginf(self.conf)      # candidates: finite or NaN scores (topk ranks NaN highest)
        k = torch.as_tensor(tokens_to_select, device=self.device).expand(self.B).to(torch.int64)
        k = torch.minimum(k, left.sum(dim=1)).to(torch.int32).reshape(self.B, 1).contiguous()
        _lib.check(self.lib.llada_topk_transfer(self.x.data_ptr(), self.x0.data_ptr(), self.conf.data_ptr(), self.B, self.L,
                                                k.data_ptr(), 1, 0, _lib.stream_ptr()))
        # committed positions stop being candidates for a further select() on the same scores
        self.conf.masked_fill_(self.x != self.mask_id, float("-inf"))
        self._block = None
        self.history[self._n_states] = (self.x.clone(), self.step_index)
        self._n_states += 1
        if self.max_history is not None:
            for s in sorted(k_ for k_ in self.history if k_ != 0)[:-self.max_history or None]:
                del self.history[s]
        return self.x

    def step(self, tokens_to_select=1, manual_selections=None):
        """forward() + select(): one denoising step."""
        self._force(manual_selections, False)       # forced tokens are part of the input of the forward
        self.forward()
        return self.select(tokens_to_select)

    def rewind(self, state):
        """Back to the tokens after the ``state``-th select() (0 = the initial, fully masked answer), and to the
        random-stream position of that moment: stepping on from here with the same arguments repeats the original
        run exactly, sampled noise included.  None if that state is unknown (never reached, or dropped by
        ``max_history``)."""
        if state not in self.history:
            return None
        x, step_index = self.history[state]
        self.x.copy_(x)
        self.step_index = step_index
        for s in [s for s in self.history if s > state]:
            del self.history[s]
        self._n_states = state + 1
        self.scores_valid = False
        self._block = None
        return self.x
