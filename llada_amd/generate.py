"""Drop-in for /root/reference/generate.py::generate -- same positional arguments and defaults
(generate.py:43-45), same return value (a fresh torch.long tensor [B, P+gen_length] on
``model.device``), same assertions and NotImplementedError.

The step loop never synchronises the host (one flag is read per block): the per-row transfer counts live in a device
schedule (get_num_transfer_tokens, generate.py:22-40, runs as a kernel once per block) and the
top-k transfer reads k from it.  With an ``llada_amd.LLaDAModel`` the forward computes logits
only for the current block's positions (all other logits are discarded by generate.py:125-128);
with any other object implementing the reference's model protocol the full ``.logits`` tensor is
consumed by the same fused kernels.
"""
import ctypes as C

import torch

from . import _lib
from .model import LLaDAModel, check_token_ids, positions_from_mask

EOS_ID = 126081   # generate.py:95,101
EOT_ID = 126348   # generate.py:101


def add_gumbel_noise(logits, temperature, *, noise=None, seed=0, step=0, row_offset=0):
    """generate.py:8-19: temperature 0 returns ``logits`` itself; otherwise float64
    ``exp(logits) / (-log u) ** temperature`` of the same shape [B, L, V].  ``noise`` (float64 uniforms, same
    shape) replaces torch.rand_like; without it u comes from the library's Philox stream, with the counters
    generate() uses for (seed, step, row_offset), so the arg-max of the result is that step's candidate token.
    generate() itself ranks these scores inside its fused epilogue and never materialises them."""
    if temperature == 0:
        return logits                                                                     # :14-15
    lib = _lib.load()
    assert logits.dim() == 3 and logits.is_cuda
    B, L, V = logits.shape
    lg = logits.to(torch.float64).contiguous()                                            # :16
    nz = None if noise is None else noise.to(device=lg.device, dtype=torch.float64).contiguous()
    assert nz is None or nz.shape == lg.shape
    out = torch.empty_like(lg)
    _lib.check(lib.llada_add_gumbel_noise(lg.data_ptr(), _lib.ptr(nz), float(temperature), out.data_ptr(), B * L, V, L,
                                          seed, step, row_offset, _lib.stream_ptr()))
    return out


def get_num_transfer_tokens(mask_index, steps):
    """generate.py:22-40 on the device: [B, n] bool -> [B, steps] int64."""
    lib = _lib.load()
    mask_index = mask_index.contiguous()
    B, n = mask_index.shape
    x = mask_index.to(torch.long).contiguous()          # 1 where masked; count tokens equal to 1
    sched = torch.empty(B, steps, dtype=torch.int32, device=x.device)
    _lib.check(lib.llada_num_transfer_tokens(x.data_ptr(), B, n, 0, n, 1, steps, sched.data_ptr(), _lib.stream_ptr()))
    return sched.to(torch.long)


@torch.no_grad()
def generate(model, prompt, attention_mask=None, steps=128, gen_length=128, block_length=128, temperature=0.,
             cfg_scale=0., remasking='low_confidence', mask_id=126336, logits_eos_inf=False,
             confidence_eos_eot_inf=False, *, seed=0, row_offset=0, noise_fn=None, rand_fn=None,
             step_callback=None, cache=None, position_ids=None):
    '''
    Args (generate.py:46-59):
        model: Mask predictor (llada_amd.LLaDAModel, or any object with .device and
               __call__(x, attention_mask=...) -> .logits).
        prompt: A tensor of shape (B, P).
        steps, gen_length, block_length, temperature, cfg_scale, remasking, mask_id,
        logits_eos_inf, confidence_eos_eot_inf: as in the reference.
    Keyword-only extensions:
        seed, row_offset: key and global row index of the library's Philox streams (Gumbel noise,
            'random' scores); data-parallel shards pass their first global row.
        noise_fn(shape) -> float64 tensor [B, L, V]: replaces torch.rand_like (generate.py:17).
        rand_fn(shape) -> float32 tensor [B, L]: replaces torch.rand (generate.py:121).
        step_callback(block, step, x, x0, confidence): called after every step (tensors are live
            device buffers; clone to keep).
        cache: None (default) recomputes every position at every step, as the reference does.
            "dual" is an APPROXIMATION outside the reference (llada_amd.LLaDAModel only): the first step
            of each block runs the whole sequence and keeps every layer's keys and values; the block's
            other steps recompute only the block's positions against those cached keys and values of the
            rest of the sequence (prefix and suffix), which are then one block stale.  Only the block's own
            positions are transfer candidates in this mode (a mask token left of the block is ignored).
            "prefix" is the milder APPROXIMATION of the same family: only the keys and values LEFT of the block
            (prompt and finished blocks) are cached at the block's first step; the block's other steps recompute
            the block and everything right of it.  Slower than "dual", closer to the reference (the still-masked
            suffix is never stale).
        position_ids: None (default) rotates every token by its column index, as the reference's model does
            whatever the padding.  "from_mask" (llada_amd.LLaDAModel only, needs ``attention_mask``) numbers
            every row from its first attended token, so that a row of a LEFT-PADDED batch with its padding
            masked is computed exactly as it is when it is run alone (up to the rounding of another GEMM
            tiling): what lets a caller batch requests the reference runs one at a time
            (eval_llada.py:260-265) and still get their batch-1 answers.  A [B, P + gen_length] integer
            tensor gives the positions directly.
    '''
    lib = _lib.load()
    device = torch.device(model.device)
    B, P = prompt.shape
    L = P + gen_length
    x = torch.full((B, L), mask_id, dtype=torch.long, device=device)                      # :60
    x[:, :P] = prompt.clone()                                                             # :61
    if attention_mask is not None:                                                        # :63-64
        attention_mask = torch.cat([attention_mask.to(device),
                                    torch.ones((B, gen_length), dtype=attention_mask.dtype, device=device)], dim=-1)
    prompt_index = (x != mask_id)                                                         # :66

    assert gen_length % block_length == 0                                                 # :68
    num_blocks = gen_length // block_length
    assert steps % num_blocks == 0                                                        # :71
    steps = steps // num_blocks
    if remasking not in _lib.REMASK:
        raise NotImplementedError(remasking)                                              # :122-123

    fast = isinstance(model, LLaDAModel)
    if fast:
        check_token_ids(prompt, model.cfg.vocab_size, mask_id, "prompt")   # the reference's embedding would raise
    if cache not in (None, "dual", "prefix"):
        raise NotImplementedError(f"cache={cache!r}")
    if cache is not None and not fast:
        raise NotImplementedError(f"cache={cache!r} needs an llada_amd.LLaDAModel")
    pos = None
    if position_ids is not None:
        if not fast or cache is not None:
            raise NotImplementedError("position_ids needs an llada_amd.LLaDAModel and cache=None")
        if isinstance(position_ids, str):
            if position_ids != "from_mask":
                raise NotImplementedError(f"position_ids={position_ids!r}")
            assert attention_mask is not None, "position_ids='from_mask' needs an attention_mask"
            pos = positions_from_mask(attention_mask)
        else:
            pos = position_ids.to(device=device, dtype=torch.int32)
            assert tuple(pos.shape) == (B, L)
        if cfg_scale > 0.0:
            pos = torch.cat([pos, pos], dim=0)
        pos = pos.contiguous()
    kv = model.kv_cache(2 * B if cfg_scale > 0.0 else B, L) if cache is not None else None
    V = model.cfg.vocab_size if fast else None
    x0 = torch.empty_like(x)
    conf = torch.empty((B, L), dtype=torch.float32, device=device)
    sched = torch.empty((B, steps), dtype=torch.int32, device=device)
    flags = (_lib.FLAG_LOGITS_EOS_INF if logits_eos_inf else 0) | (_lib.FLAG_CONF_EOS_EOT_INF if confidence_eos_eot_inf else 0)
    stream = _lib.stream_ptr()
    am2 = None
    if cfg_scale > 0.0:
        if attention_mask is None:
            # generate.py:86-88 binds attention_mask_ only when a mask was given
            raise UnboundLocalError("cannot access local variable 'attention_mask_' where it is not associated with a value")
        am2 = torch.cat([attention_mask, attention_mask], dim=0)
    batch_idx = torch.arange(B, device=device, dtype=torch.int32)[:, None] * L
    # scratch of the streamed row statistics (temperature 0, confidence rules): one allocation per call
    ws_rows = B * L                                   # B * block_length rows in the usual case
    ws = torch.empty(lib.llada_rows_workspace_bytes(ws_rows), dtype=torch.uint8, device=device)

    for num_block in range(num_blocks):                                                   # :74
        bs, be = P + num_block * block_length, P + (num_block + 1) * block_length
        _lib.check(lib.llada_num_transfer_tokens(x.data_ptr(), B, L, bs, be, mask_id, steps, sched.data_ptr(), stream))
        # Logits are needed where a token can be committed: mask tokens left of block_end (:125-128 exclude only
        # the positions after the block).  Normally those are the block's own positions; a mask token inside the
        # prompt, or a position of an earlier block for which the model predicted the mask token itself, is a
        # candidate too in the reference -- one flag read per block decides (the step loop stays free of host reads).
        row_lo = bs
        if fast and kv is None and bs > 0 and bool((x[:, :bs] == mask_id).any().item()):   # cache='dual': block rows only
            row_lo = 0
        rows = (batch_idx + torch.arange(row_lo, be, device=device, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        n_rows = rows.numel()
        rows2 = torch.cat([rows, rows + B * L]).contiguous() if cfg_scale > 0.0 else None
        # cached steps recompute [bs, ce) of every sequence: the block ("dual") or everything from the block on
        # ("prefix"); logits are asked for the block's positions only
        ce = be if cache == "dual" else L
        crows = None
        if cache == "prefix":
            nb_ = 2 * B if cfg_scale > 0.0 else B
            crows = (torch.arange(nb_, device=device, dtype=torch.int32)[:, None] * (ce - bs)
                     + torch.arange(0, be - bs, device=device, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        logits_buf = None
        if fast:
            logits_buf = torch.empty((2 * n_rows if cfg_scale > 0.0 else n_rows, V), dtype=torch.bfloat16, device=device)
        for i in range(steps):                                                            # :80
            un_ptr = None
            if cfg_scale > 0.0:                                                           # :82-90
                un_x = torch.where(prompt_index, mask_id, x)
                x_ = torch.cat([x, un_x], dim=0)
                if fast:
                    if kv is not None and i > 0:
                        lg = model.forward_block(x_[:, bs:ce], am2, L, bs, kv, out=logits_buf, update_cache=False,
                                                 logit_rows=crows)
                    else:
                        lg = model.forward_rows(x_, am2, rows2, out=logits_buf, kv_cache=kv, position_ids=pos)
                    cond, un = lg[:n_rows], lg[n_rows:]
                    row_pos, nr = rows, n_rows
                else:
                    lg = model(x_, attention_mask=am2).logits
                    lg = lg.to(device=device, dtype=torch.bfloat16).contiguous()
                    cond, un = lg[:B].reshape(B * L, -1), lg[B:].reshape(B * L, -1)
                    row_pos, nr = None, B * L
                un_ptr = un.data_ptr()
            else:
                if fast and kv is not None and i > 0:
                    cond = model.forward_block(x[:, bs:ce], attention_mask, L, bs, kv, out=logits_buf, update_cache=False,
                                               logit_rows=crows)
                    row_pos, nr = rows, n_rows
                elif fast:
                    cond = model.forward_rows(x, attention_mask, rows, out=logits_buf, kv_cache=kv, position_ids=pos)   # :92
                    row_pos, nr = rows, n_rows
                else:
                    lg = model(x, attention_mask=attention_mask).logits
                    cond = lg.to(device=device, dtype=torch.bfloat16).contiguous().reshape(B * L, -1)
                    row_pos, nr = None, B * L
            vocab = cond.shape[-1]
            gstep = num_block * steps + i
            noise = rand_scores = None
            if temperature != 0 and noise_fn is not None:
                noise = noise_fn((B, L, vocab)).to(device=device, dtype=torch.float64).reshape(B * L, vocab)
                if row_pos is not None:
                    noise = noise.index_select(0, row_pos.long())
                noise = noise.contiguous()
            if remasking == 'random' and rand_fn is not None:
                rand_scores = rand_fn((B, L)).to(device=device, dtype=torch.float32).contiguous()
            args = _lib.RemaskArgs(cond.data_ptr(), un_ptr, float(cfg_scale), _lib.ptr(row_pos), nr, vocab,
                                   x.data_ptr(), B, L, be, mask_id, _lib.REMASK[remasking], flags, EOS_ID, EOT_ID,
                                   float(temperature), _lib.ptr(noise), _lib.ptr(rand_scores), seed, gstep,
                                   row_offset, x0.data_ptr(), conf.data_ptr(), ws.data_ptr(), ws.numel())
            _lib.check(lib.llada_remask_scores(C.byref(args), stream))                    # :94-128
            _lib.check(lib.llada_topk_transfer(x.data_ptr(), x0.data_ptr(), conf.data_ptr(), B, L,
                                               sched.data_ptr(), steps, i, stream))       # :130-134
            if step_callback is not None:
                step_callback(num_block, i, x, x0, conf)
    return x                                                                              # :136
