"""Data-parallel use of one 8 x MI355X node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), weights replicated, work split along the path's natural axes.

  * generate: prompt rows are independent (per-row schedule generate.py:30-38, per-row top-k
    :130-133), so each rank samples a contiguous shard of rows and ONE all-gather of the
    [rows, L] int64 outputs ends the call.  Random streams are keyed by the global row index,
    so the draws do not depend on the number of ranks (the forward's GEMM tiling does depend on
    the rows per GPU, i.e. logits agree across rank counts to bf16 rounding noise, not bit for bit).
  * get_log_likelihood: every iteration's batch rows are dealt over ranks inside
    ``llada_amd.get_log_likelihood`` (rank/world_size arguments) and the per-iteration loss
    sums are all-reduced once.
Nothing is communicated inside the denoising loop.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size):
    """Contiguous, balanced row ranges: rank r owns [bounds[r], bounds[r+1])."""
    base, rem = divmod(n_rows, world_size)
    bounds = [0]
    for r in range(world_size):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def generate_data_parallel(generate_fn, model, prompt, attention_mask=None, *args, process_group=None, **kwargs):
    """Run ``generate_fn`` (llada_amd.generate) on this rank's shard of the prompt rows and
    all-gather the outputs; every rank returns the full [B, P+gen_length] tensor."""
    if not (dist.is_available() and dist.is_initialized()):
        return generate_fn(model, prompt, attention_mask, *args, **kwargs)
    world = dist.get_world_size(process_group)
    rank = dist.get_rank(process_group)
    B = prompt.shape[0]
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    max_rows = max(bounds[r + 1] - bounds[r] for r in range(world))
    local = None
    if hi > lo:
        am = None if attention_mask is None else attention_mask[lo:hi]
        local = generate_fn(model, prompt[lo:hi], am, *args, row_offset=lo, **kwargs)
    # equal-size buffers for the all-gather (ranks with fewer rows pad)
    if local is not None:
        L = local.shape[1]
        device, dtype = local.device, local.dtype
    else:
        gen_length = kwargs.get("gen_length", args[1] if len(args) > 1 else 128)
        L = prompt.shape[1] + gen_length
        device, dtype = torch.device(model.device), torch.long
    buf = torch.zeros((max_rows, L), dtype=dtype, device=device)
    if local is not None:
        buf[: hi - lo] = local
    on_host = buf.is_cuda and dist.get_backend(process_group) == "gloo"   # a CPU-only process group (rehearsals)
    if on_host:
        buf = buf.cpu()
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=process_group)
    out = torch.cat([gathered[r][: bounds[r + 1] - bounds[r]] for r in range(world)], dim=0)
    return out.to(device) if on_host else out


def combine_loss_sums(loss_sums, batch_size, world_size=1, process_group=None):
    """get_log_likelihood.py:73-77 across ranks: ``loss_sums`` [n_iter] float64 holds this rank's
    sum of CE/p_mask per iteration; one all-reduce(sum), then loss = sum / batch_size per
    iteration and the result is minus their mean.  The only host synchronisation of the call.
    The all-reduce also runs for a one-rank group when a ``process_group`` is named (a no-op numerically; it keeps
    the single-rank call on the same code path as the sharded one)."""
    if world_size > 1 or process_group is not None:
        if loss_sums.is_cuda and dist.get_backend(process_group) == "gloo":
            loss_sums = loss_sums.cpu()        # a CPU-only process group (rehearsals): reduce on the host
        dist.all_reduce(loss_sums, op=dist.ReduceOp.SUM, group=process_group)
    loss_ = (loss_sums / batch_size).cpu().tolist()
    return - sum(loss_) / len(loss_)
