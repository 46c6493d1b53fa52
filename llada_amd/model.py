"""The mask predictor object: the model protocol the reference samplers use
(``model.device`` at generate.py:60,64; ``model(x, attention_mask=...).logits`` at generate.py:88,92 and
get_log_likelihood.py:38) on top of the HIP forward, plus the row-restricted fast path that
``llada_amd.generate`` / ``llada_amd.get_log_likelihood`` use (LM head only on the rows that
can influence the result).  PyTorch only owns the memory and the stream."""
import ctypes as C

import torch

from . import _lib
from .weights import LLaDAConfig, LLaDAWeights


class ModelOutput:
    """Stand-in for the HF output object: only ``.logits`` is read by the reference."""

    def __init__(self, logits):
        self.logits = logits


def check_token_ids(ids, vocab_size, mask_id=None, what="input_ids"):
    """The reference's embedding lookup and F.cross_entropy raise for ids outside [0, vocab) (an IndexError from
    torch); the kernels would clamp them instead, so the drop-ins check once per call, outside the step loop
    (one host read)."""
    if mask_id is not None and not (0 <= int(mask_id) < vocab_size):
        raise IndexError(f"mask_id {mask_id} is outside the vocabulary of {vocab_size} entries")
    if ids.numel() == 0:
        return
    lo, hi = torch.aminmax(ids)
    lo, hi = int(lo.item()), int(hi.item())
    if lo < 0 or hi >= vocab_size:
        raise IndexError(f"{what}: token id {lo if lo < 0 else hi} is outside the vocabulary of {vocab_size} entries "
                         "(index out of range)")


def positions_from_mask(attention_mask):
    """Rotary positions that count from every row's first attended token: cumsum(mask) - 1, clamped at 0 on the
    left padding.  With them (and the mask over the padding) a row of a left-padded batch is rotated exactly as it
    is when it runs alone; the reference's model numbers by column whatever the padding (rotary attention depends
    on position differences only, so the two agree up to rounding)."""
    m = (attention_mask != 0).to(torch.int32)
    return (torch.cumsum(m, dim=-1) - 1).clamp_(min=0).to(torch.int32)


class LLaDAModel:
    def __init__(self, weights: LLaDAWeights):
        self.lib = _lib.load()
        self.weights = weights
        self.cfg: LLaDAConfig = weights.cfg
        self.device = weights.device
        c = self.cfg
        self._c_cfg = _lib.Config(c.d_model, c.n_layers, c.n_heads, c.mlp_hidden, c.vocab_size, c.rms_eps)

        def arr(ts):
            return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

        w = weights
        self._arrays = [arr(w.attn_norm), arr(w.wqkv), arr(w.wo), arr(w.ff_norm), arr(w.wgu), arr(w.wdown)]
        self._c_w = _lib.Weights((1 if w.packed else 0) | (2 if w.tiled else 0), w.wte.data_ptr(), w.ln_f.data_ptr(), w.lm_head.data_ptr(),
                                 *[C.cast(a, C.POINTER(C.c_void_p)) for a in self._arrays])
        self._ws = None
        self._ws_block = None
        self._rope = {}

    # -- HF-like surface the reference's callers touch -------------------------------------------
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    @property
    def dtype(self):
        return torch.bfloat16

    # -- helpers -----------------------------------------------------------------------------------
    def rope_tables(self, L):
        if L not in self._rope:
            c = self.cfg
            inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))
            ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
            self._rope[L] = (torch.cos(ang).to(self.device).contiguous(), torch.sin(ang).to(self.device).contiguous())
        return self._rope[L]

    def _scratch(self, attr, need):
        """The cached scratch buffer `attr`, grown when too small.  The buffer outlives the call that allocated it
        and is handed to kernels as a raw pointer, so the caching allocator is told about every stream it is
        used on (a caller may switch streams between calls); the old buffer of a grow is released only to
        streams that have been told the same."""
        ws = getattr(self, attr)
        stream = torch.cuda.current_stream(self.device)
        if ws is None or ws.numel() < need:
            if ws is not None:
                ws.record_stream(stream)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            setattr(self, attr, ws)
        ws.record_stream(stream)
        return ws

    def _workspace(self, n_seq, L, n_rows):
        return self._scratch("_ws", self.lib.llada_forward_workspace_bytes(C.byref(self._c_cfg), n_seq, L, n_rows))

    def kv_cache(self, n_seq, L):
        """Storage for the block-cached sampler: every layer's [Q | K | V] image of n_seq x L tokens."""
        need = self.lib.llada_kv_cache_bytes(C.byref(self._c_cfg), n_seq, L)
        return torch.empty(need, dtype=torch.uint8, device=self.device)

    def forward_block(self, block_ids, attention_mask, seq_len, block_start, kv_cache, out=None, update_cache=True,
                      logit_rows=None):
        """Recompute only the positions [block_start, block_start + block_len) of every sequence
        (block_ids [n_seq, block_len]) against ``kv_cache`` (filled by forward_rows(..., kv_cache=)): their
        keys and values replace the cached ones (``update_cache=False``: they are used for this call only and
        the cache stays as it is; same logits).  ``attention_mask`` is the full [n_seq, seq_len] mask.
        Returns bf16 logits [n_seq * block_len, V], or [n_rows, V] for the rows ``logit_rows`` (int32 indices
        seq * block_len + offset into the recomputed rows)."""
        assert block_ids.dim() == 2 and block_ids.dtype == torch.long
        ids = block_ids.to(self.device).contiguous()
        n_seq, bl = ids.shape
        km = None
        if attention_mask is not None:
            km = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
            assert tuple(km.shape) == (n_seq, seq_len)
        assert kv_cache.numel() >= self.lib.llada_kv_cache_bytes(C.byref(self._c_cfg), n_seq, seq_len)
        cos, sin = self.rope_tables(seq_len)
        n_rows = n_seq * bl if logit_rows is None else int(logit_rows.numel())
        if logit_rows is not None:
            assert logit_rows.dtype == torch.int32 and logit_rows.is_contiguous() and logit_rows.device == ids.device
        if out is None:
            out = torch.empty(n_rows, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device)
        ws = self._scratch("_ws_block", self.lib.llada_forward_block_workspace_bytes(C.byref(self._c_cfg), n_seq, seq_len, bl))
        _lib.check(self.lib.llada_forward_block(C.byref(self._c_cfg), C.byref(self._c_w), ids.data_ptr(), _lib.ptr(km),
                                                cos.data_ptr(), sin.data_ptr(), n_seq, seq_len, block_start, bl,
                                                kv_cache.data_ptr(), 1 if update_cache else 0, _lib.ptr(logit_rows),
                                                n_rows if logit_rows is not None else 0, out.data_ptr(),
                                                ws.data_ptr(), ws.numel(),
                                                _lib.stream_ptr()))
        return out

    def forward_rows(self, input_ids, attention_mask=None, logit_rows=None, out=None, kv_cache=None, position_ids=None):
        """Forward on ids [n_seq, L]; returns bf16 logits [n_rows, V] for the flat token indices in
        ``logit_rows`` (int32 tensor) or for every token when it is None.  With ``kv_cache`` every layer's
        [Q | K | V] image is left there for forward_block.  ``position_ids`` [n_seq, L] (int): the rotary position
        of every token instead of its column index (see positions_from_mask); not together with ``kv_cache``."""
        assert input_ids.dim() == 2 and input_ids.dtype == torch.long
        ids = input_ids.to(self.device).contiguous()
        n_seq, L = ids.shape
        n_rows = n_seq * L if logit_rows is None else int(logit_rows.numel())
        if logit_rows is not None:
            assert logit_rows.dtype == torch.int32 and logit_rows.is_contiguous() and logit_rows.device == ids.device
        km = None
        if attention_mask is not None:
            km = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
            assert km.shape == ids.shape
        cos, sin = self.rope_tables(L)
        if out is None:
            out = torch.empty(n_rows, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device)
        ws = self._workspace(n_seq, L, n_rows)
        if position_ids is not None:
            if kv_cache is not None:
                raise NotImplementedError("position_ids with a kv_cache")
            pos = position_ids.to(device=self.device, dtype=torch.int32).contiguous()
            assert tuple(pos.shape) == (n_seq, L)
            rc = self.lib.llada_forward_positions(C.byref(self._c_cfg), C.byref(self._c_w), ids.data_ptr(), _lib.ptr(km),
                                                  pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), n_seq, L,
                                                  _lib.ptr(logit_rows), n_rows, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _lib.stream_ptr())
        elif kv_cache is None:
            rc = self.lib.llada_forward(C.byref(self._c_cfg), C.byref(self._c_w), ids.data_ptr(), _lib.ptr(km),
                                        cos.data_ptr(), sin.data_ptr(), n_seq, L, _lib.ptr(logit_rows), n_rows,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        else:
            assert kv_cache.numel() >= self.lib.llada_kv_cache_bytes(C.byref(self._c_cfg), n_seq, L)
            rc = self.lib.llada_forward_fill_cache(C.byref(self._c_cfg), C.byref(self._c_w), ids.data_ptr(), _lib.ptr(km),
                                                   cos.data_ptr(), sin.data_ptr(), n_seq, L, _lib.ptr(logit_rows), n_rows,
                                                   out.data_ptr(), ws.data_ptr(), ws.numel(), kv_cache.data_ptr(),
                                                   _lib.stream_ptr())
        _lib.check(rc)
        return out

    def residual_stream(self, input_ids, attention_mask=None, n_layers=None):
        """The residual stream [n_seq, L, d] (bf16) after the first ``n_layers`` blocks (default: all), before the
        final norm -- the per-layer ``hidden_states`` of an HF model.  Runs the forward truncated to that depth and
        copies the stream out of the workspace (llada_forward leaves it in the first n_seq * L * d elements)."""
        ids = input_ids.to(self.device).contiguous()
        n_seq, L = ids.shape
        nl = self.cfg.n_layers if n_layers is None else int(n_layers)
        assert 0 <= nl <= self.cfg.n_layers
        c = self.cfg
        cc = _lib.Config(c.d_model, nl, c.n_heads, c.mlp_hidden, c.vocab_size, c.rms_eps)
        km = None
        if attention_mask is not None:
            km = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
        cos, sin = self.rope_tables(L)
        M = n_seq * L
        # every row is asked for, so the last layer runs on all of them (no row pruning)
        out = torch.empty(M, c.vocab_size, dtype=torch.bfloat16, device=self.device)
        ws = self._scratch("_ws", self.lib.llada_forward_workspace_bytes(C.byref(cc), n_seq, L, M))
        _lib.check(self.lib.llada_forward(C.byref(cc), C.byref(self._c_w), ids.data_ptr(), _lib.ptr(km), cos.data_ptr(),
                                          sin.data_ptr(), n_seq, L, None, M, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr()))
        return ws[:n_seq * L * c.d_model * 2].view(torch.bfloat16).view(n_seq, L, c.d_model).clone()

    def __call__(self, input_ids, attention_mask=None, position_ids=None):
        n_seq, L = input_ids.shape
        logits = self.forward_rows(input_ids, attention_mask, None, position_ids=position_ids)
        return ModelOutput(logits.view(n_seq, L, self.cfg.vocab_size))

    forward = __call__
