"""bench.py host logic that needs no GPU: the self-launching of ranks (the parent starts one child per rank
before anything touches the GPU and fails when a rank fails) and the torch-CPU restatement of the forward that
the `cpu_baseline` leg times (oracle/model_torch.py) against the numpy oracle."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_and_propagates_failure():
    """Without a launcher `--gpus 2` must spawn two ranks itself.  In a container without a GPU every rank exits
    with an error before any device work: the parent's status must be non-zero and both ranks must have been
    started with RANK / WORLD_SIZE set (each says which rank it is)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LLADA_BENCH_ECHO_RANK"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return      # on a GPU box the GPU tests cover the launch
    assert res.returncode != 0
    assert "bench rank 0 of 2" in res.stderr and "bench rank 1 of 2" in res.stderr


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr


def test_torch_cpu_forward_matches_numpy_oracle():
    import torch

    from oracle import model_torch as mt
    from oracle.model import LLaDAConfig, forward, init_weights
    cfg = LLaDAConfig(d_model=256, n_layers=2, n_heads=2, mlp_hidden=512, vocab_size=512)
    W = init_weights(cfg, seed=3, std=0.05, head_scale=4.0)
    ids = np.random.default_rng(0).integers(0, 512, size=(2, 24))
    am = np.ones((2, 24), dtype=np.int64)
    am[0, :5] = 0
    Wt = mt.from_numpy_weights(W)
    W32 = {k: v.float() for k, v in Wt.items()}
    for mask in (None, am):
        ref = forward(W, cfg, ids, mask, True)
        for mode, weights in (("f32", Wt), ("f32", W32), ("bf16", Wt)):
            out = mt.forward(weights, cfg, torch.from_numpy(ids), None if mask is None else torch.from_numpy(mask), mode)
            out = out.float().numpy()
            assert np.abs(out - ref).max() <= 2e-2 * np.abs(ref).max(), mode
    sub = mt.forward(Wt, cfg, torch.from_numpy(ids), None, "f32", logit_positions=[3, 7]).float().numpy()
    full = mt.forward(Wt, cfg, torch.from_numpy(ids), None, "f32").float().numpy()
    assert np.array_equal(sub, full[:, [3, 7]])
    assert mt.pick_mode(d=256, rows=16) in ("bf16", "f32")


def test_clock_power_sampler_is_harmless_without_the_tools(monkeypatch, tmp_path):
    """The `under_load` fields of the bench record come from rocm-smi / amd-smi; on a machine without them (or
    with tools that print something else) the sampler must leave the record alone, never raise."""
    sys.path.insert(0, ROOT)
    import bench
    out = {}
    monkeypatch.setenv("PATH", str(tmp_path))          # no vendor tools on the path
    bench.sample_clock_power(0.0, out)
    assert out == {}
    for name, text in (("rocm-smi", "GPU[0]\t\t: sclk clock level: 1: (1803Mhz)"), ("amd-smi", "        SOCKET_POWER: 1398 W")):
        tool = tmp_path / name
        tool.write_text("#!/bin/sh\necho '%s'\n" % text)
        tool.chmod(0o755)
    bench.sample_clock_power(0.0, out)
    assert out == {"sclk_mhz": 1803, "socket_power_w": 1398}
