"""world_size-2 gloo runs (CPU) of the data-parallel wrappers: row sharding + all-gather for
generate, row sharding + all-reduce for the likelihood.  The per-rank compute is played by the
oracle (the HIP kernels need a GPU); what is under test is the partition/collective logic, which is
the same code the GPU ranks run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Model:
    device = "cpu"


def _row_keyed_generate(model, prompt, attention_mask=None, steps=4, gen_length=8, block_length=4, *, row_offset=0,
                        **kw):
    """Per-row deterministic stand-in for llada_amd.generate: the 'sampled' tokens depend on the
    GLOBAL row index, like the library's Philox streams do."""
    B, P = prompt.shape
    out = torch.zeros((B, P + gen_length), dtype=torch.long)
    out[:, :P] = prompt
    for r in range(B):
        g = np.random.Generator(np.random.PCG64(1000 + row_offset + r))
        out[r, P:] = torch.from_numpy(g.integers(0, 100, size=gen_length))
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llada_amd.parallel import combine_loss_sums, generate_data_parallel, shard_bounds
    from oracle import likelihood as oll
    from tests.golden_util import FakeModel
    res = {}
    # ---- generate: B = 5 rows over 2 ranks (uneven shards) ----
    prompt = torch.arange(5 * 3).reshape(5, 3)
    full = generate_data_parallel(_row_keyed_generate, _Model(), prompt, None, 4, 8, 4)
    res["gen"] = full.numpy()
    # ---- likelihood: every iteration's rows dealt over ranks, one all-reduce at the end ----
    b, l1, l2, V, n_iter = 6, 4, 9, 64, 3
    g = np.random.Generator(np.random.PCG64(5))
    seqrow = g.integers(0, V, size=l1 + l2)
    seq = np.repeat(seqrow[None], b, axis=0)
    bounds = shard_bounds(b, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    sums = torch.zeros(n_iter, dtype=torch.float64)
    for it in range(n_iter):
        dg = np.random.Generator(np.random.PCG64(100 + it))
        k = int(dg.integers(1, l2 + 1))
        perms = np.stack([dg.permutation(l2) for _ in range(b)])
        noisy, p_mask = oll.forward_process(seq, l1, 999, k, perms)
        logits = FakeModel(7 + it, V)(seq)
        m = noisy[lo:hi] == 999
        ce = oll.masked_ce(logits[lo:hi][m], seq[lo:hi][m])
        sums[it] = float((ce / p_mask[lo:hi][m]).sum(dtype=np.float64))
    res["ll"] = combine_loss_sums(sums, b, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res))


@pytest.mark.timeout(180)
def test_data_parallel_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answers
    sys.path.insert(0, ROOT)
    from oracle import likelihood as oll
    from tests.golden_util import FakeModel
    prompt = torch.arange(5 * 3).reshape(5, 3)
    want = _row_keyed_generate(_Model(), prompt, None, 4, 8, 4).numpy()
    for r in range(2):
        assert np.array_equal(got[r]["gen"], want)
    b, l1, l2, V, n_iter = 6, 4, 9, 64, 3
    g = np.random.Generator(np.random.PCG64(5))
    seq = np.repeat(g.integers(0, V, size=l1 + l2)[None], b, axis=0)
    losses = []
    for it in range(n_iter):
        dg = np.random.Generator(np.random.PCG64(100 + it))
        k = int(dg.integers(1, l2 + 1))
        perms = np.stack([dg.permutation(l2) for _ in range(b)])
        noisy, p_mask = oll.forward_process(seq, l1, 999, k, perms)
        logits = FakeModel(7 + it, V)(seq)
        m = noisy == 999
        ce = oll.masked_ce(logits[m], seq[m])
        losses.append(float((ce / p_mask[m]).sum(dtype=np.float64)) / b)
    single = -sum(losses) / len(losses)
    assert abs(got[0]["ll"] - single) <= 1e-12 * abs(single)
    assert got[0]["ll"] == got[1]["ll"]
