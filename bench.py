#!/usr/bin/env python3
"""Headline benchmark: LLaDA-8B generate(), gen_length=256, steps=256, block_length=32, synthetic
prompts (P=128) and random-init weights at the true shapes, data-parallel over N GPUs with a fixed
number of prompt rows per GPU (weak scaling; 8 rows/GPU = the 64-prompt north-star batch on 8 GPUs).

One "step" = one denoising step of the whole per-GPU batch: the non-causal forward over all
B*(P+G) tokens plus the fused remasking epilogue.  `value` = generated tokens/s of the whole job =
N*B*G / (steps_total * time_per_step).

`python bench.py --gpus N` launches its own N ranks (one process per GPU, RCCL) when no launcher has set
WORLD_SIZE; under `python -m torch.distributed.run` it uses the ranks it is given.  `--total-batch T` switches to
strong scaling: T prompts in all, dealt over the ranks (the 64-prompt strong-scaling configuration is `--total-batch 64 --gen 512
--block 64`).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel = the gate/up projection GEMM (SwiGLU epilogue); achieved = its
                  algorithmic FLOPs per launch / its mean launch duration measured with HIP events
                  inside the timed region; peak = 2.5 PFLOP/s dense bf16.
  cpu_baseline -- the numpy oracle timed on this host on a bounded sample of the same step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, seconds_budget=25.0):
    """The reference's CPU path (the CPU-runnable benchmark configuration: batch 1, gen_length 64, steps 64, block_length 32,
    temperature 0, low_confidence, P = 128 -> 192 tokens per forward) timed on this host: oracle/model_torch.py
    (torch-CPU restatement of the forward on all the threads torch has) + the oracle's numpy epilogue, full
    logits for every position as the reference computes them (generate.py:92).  Bounded sample: the first few
    denoising steps of the 64 (every step costs the same: no cache, fixed length), 32 layers per step cycling 4
    distinct random weight sets (1.7 GB in bf16, far beyond any CPU cache) so that the host need not hold 15 GB.
    kind = "port": the reference's own model code is not in its tree."""
    import numpy as np
    import torch

    from oracle import model_torch as mt
    from oracle import sampler as osamp
    from oracle.model import LLaDAConfig as OCfg

    threads = torch.get_num_threads()
    mode = mt.pick_mode()
    d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
    n_sets = 4
    ocfg = OCfg(d_model=d, n_layers=cfg.n_layers, n_heads=cfg.n_heads, mlp_hidden=F, vocab_size=V)
    wdt = torch.bfloat16 if mode == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(0)
    base = (torch.randn(F, d, generator=gen) * 0.02)

    def w(n, k, shift, scale=1.0):
        # timing only: every matrix is a rolled view of one random base (values N(0, 0.02^2), all distinct memory)
        t = torch.roll(base, shifts=shift, dims=0)
        t = t[:n, :k] if k <= d else t.t()[:n, :k]
        return (t * scale).to(wdt).contiguous()

    W = {"wte": w(4096, d, 1).to(torch.bfloat16), "ln_f": torch.ones(d), "ff_out": torch.cat([w(F, d, 3 + j, 8.0) for j in range(-(-V // F))])[:V].contiguous()}
    for i in range(n_sets):
        p = f"blocks.{i}."
        for j, (name, shape) in enumerate((("q_proj", (d, d)), ("k_proj", (d, d)), ("v_proj", (d, d)), ("attn_out", (d, d)),
                                           ("ff_proj", (F, d)), ("up_proj", (F, d)), ("ff_out", (d, F)))):
            W[p + name] = w(*shape, 17 * i + j + 20)
        W[p + "attn_norm"] = torch.ones(d)
        W[p + "ff_norm"] = torch.ones(d)
    P, G, bl, mask_id = 128, 64, 32, cfg.mask_token_id
    g = np.random.Generator(np.random.PCG64(1234))
    x = np.full((1, P + G), mask_id, dtype=np.int64)
    x[:, :P] = g.integers(0, 126000, size=(1, P))
    times = []
    t_all = time.perf_counter()
    step = 0
    while step < G // 2 and (step < 2 or time.perf_counter() - t_all < seconds_budget):   # steps of the first block
        t0 = time.perf_counter()
        logits = mt.forward(W, ocfg, torch.from_numpy(x), None, mode, layer_of=lambda i: i % n_sets)      # generate.py:92
        x0, conf = osamp.confidence_step(x, logits.float().numpy(), mask_id, P + bl, "low_confidence")   # :97-128
        osamp.topk_transfer(x, x0, conf, np.ones(1, dtype=np.int64))                                     # :130-134
        times.append(time.perf_counter() - t0)
        step += 1
    t_step = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"value": round(1.0 / t_step, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": os.cpu_count(), "gemm_mode": mode,
            "ms_per_step": round(1e3 * t_step, 1),
            "sample": (f"BASELINE config 1 (batch 1, P=128, gen_length=64, steps=64, block_length=32, T=0, low_confidence): "
                       f"{len(times)} of its 64 denoising steps, each one 32-layer torch-CPU forward over 192 tokens "
                       f"({mode} GEMMs, {threads} threads, 4 weight sets cycled) + full-vocabulary numpy epilogue; "
                       f"median step {t_step:.2f} s; 1 token per step")}


def sample_clock_power(delay_s, out):
    """Shader clock and socket power as the vendor tools report them, read once `delay_s` seconds from now (called on a
    thread while a workload runs on the GPU).  Informational: the record carries them next to the roofline so that the
    2.4 GHz peak can be compared with the clock the chip actually holds; every failure leaves the fields out."""
    import re
    import subprocess
    time.sleep(delay_s)
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        if m:
            out["sclk_mhz"] = int(m.group(1))
    except Exception:
        pass
    try:
        txt = subprocess.run(["amd-smi", "metric", "--power"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"SOCKET_POWER:\s*(\d+)\s*W", txt)
        if m:
            out["socket_power_w"] = int(m.group(1))
    except Exception:
        pass


def measured_ceilings(torch, device, M, N, K):
    """Yardsticks measured on this GPU next to the vendor peaks: the vendor library's bf16 GEMM
    (torch.matmul -> hipBLASLt) at the dominant kernel's shape, and a device-to-device copy.
    Neither is part of the product path."""
    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    a = torch.randn(M, K, device=device).to(torch.bfloat16)
    w = (torch.randn(N, K, device=device) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    t_gemm = timed(lambda: torch.matmul(a, w.t(), out=out), 20)
    src = torch.empty(1 << 30, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    t_copy = timed(lambda: dst.copy_(src), 10)
    return {"vendor_gemm_tflops": round(2.0 * M * N * K / t_gemm / 1e12, 1), "vendor_gemm_shape": [M, N, K],
            "memcpy_tb_s": round(2.0 * (1 << 30) / t_copy / 1e12, 2)}


class _TimedWindowDone(Exception):
    pass


def window_start(W_, K_, steps_per_block, steps_total):
    """First timed step: W_ untimed steps, plus as many more as it takes for the K_ timed steps to contain a block
    boundary (the per-block work -- schedule kernel, one flag read, buffer set-up -- then falls inside the window, as
    it does in a whole call), when the call is long enough for that."""
    if steps_per_block >= steps_total or (W_ // steps_per_block) != ((W_ + K_ - 1) // steps_per_block):
        return W_
    start = (W_ // steps_per_block + 1) * steps_per_block - K_ // 2
    return start if start >= W_ and start + K_ <= steps_total else W_


def event_timed_steps(torch, generate, model, prompt, G, bl, W_, K_, **sampler):
    """Seconds of GPU time for K_ consecutive steps of ONE uninterrupted generate() call (steps == gen_length), between
    two HIP events recorded on the launch stream after step W_ and after step W_ + K_: no host synchronisation inside
    the window, so the device never idles waiting for the host to refill its queue (a synchronize right before a short
    window of 7 ms steps costs the window several ms of restart: queue refill and clock ramp)."""
    kw = dict(temperature=0., cfg_scale=0., remasking="low_confidence")
    kw.update(sampler)
    W_ = window_start(W_, K_, bl, G)
    ev = {}

    def cb(block, step, x, x0, conf):
        n = block * bl + step + 1
        if n in (W_, W_ + K_):
            ev[n] = torch.cuda.Event(enable_timing=True)
            ev[n].record()
            if n == W_ + K_:
                raise _TimedWindowDone

    try:
        generate(model, prompt, None, steps=G, gen_length=G, block_length=bl, step_callback=cb, **kw)
    except _TimedWindowDone:
        pass
    torch.cuda.synchronize()
    return ev[W_].elapsed_time(ev[W_ + K_]) * 1e-3


def timed_steps(generate, model, prompt, G, bl, W_, K_, sync_all, on_start=None, row_offset=0, attention_mask=None,
                span_block=False, **sampler):
    """Runs generate() (steps == gen_length) and times K_ consecutive steps of it, after at least W_ untimed ones,
    between two barrier + synchronize pairs; returns seconds for the K_ steps.  ``span_block``: place the window
    over a block boundary (window_start).  ``sampler``: temperature, cfg_scale, remasking, seed of the call
    (default: temperature 0, no guidance, low_confidence)."""
    kw = dict(temperature=0., cfg_scale=0., remasking="low_confidence")
    kw.update(sampler)
    state = {"n": 0, "t0": None, "t1": None}
    if span_block:
        W_ = window_start(W_, K_, bl, G)

    def begin():
        sync_all()
        if on_start is not None:
            on_start()
        state["t0"] = time.perf_counter()

    def cb(block, step, x, x0, conf):
        state["n"] += 1
        if state["n"] == W_:
            begin()
        elif state["n"] == W_ + K_:
            sync_all()
            state["t1"] = time.perf_counter()
            raise _TimedWindowDone          # the rest of the call is not needed (64 rows x 512 steps take minutes)

    if W_ == 0:
        begin()
    while state["t1"] is None:
        try:
            generate(model, prompt, attention_mask, steps=G, gen_length=G, block_length=bl, row_offset=row_offset,
                     step_callback=cb, **kw)
        except _TimedWindowDone:
            pass
    return state["t1"] - state["t0"]


def epilogue_roofline(torch, lib, _lib, device, rows, V, mask_id):
    """The fused remasking epilogue alone (HBM-bound): `rows` full-vocabulary logits rows per launch,
    timed with events on the launch stream over a ring of buffers larger than the caches.  The step's
    own launch (rows = prompts x block_length) is short enough that the fixed cost of its two kernels
    shows; `large_launch` times the same entry point on 2048 rows, where the streaming rate shows."""
    st = torch.cuda.current_stream().cuda_stream

    def timed(n_rows):
        n_ring = max(2, min(8, int(1.1e9 // (n_rows * V * 2)) + 1))
        ring = [torch.randn(n_rows, V, device=device).mul_(3).to(torch.bfloat16) for _ in range(n_ring)]
        x = torch.full((1, n_rows), mask_id, dtype=torch.long, device=device)
        x0 = torch.empty_like(x)
        conf = torch.empty((1, n_rows), dtype=torch.float32, device=device)
        ws = torch.empty(lib.llada_rows_workspace_bytes(n_rows), dtype=torch.uint8, device=device)

        def call(i):
            a = _lib.RemaskArgs(ring[i % n_ring].data_ptr(), None, 0.0, None, n_rows, V, x.data_ptr(), 1, n_rows, n_rows,
                                mask_id, _lib.REMASK["low_confidence"], 0, 126081, 126348, 0.0, None, None, 0, 0, 0,
                                x0.data_ptr(), conf.data_ptr(), ws.data_ptr(), ws.numel())
            _lib.check(lib.llada_remask_scores(C.byref(a), st))
        for i in range(4):
            call(i)
        torch.cuda.synchronize()
        n = 32
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    t = timed(rows)
    nbytes = rows * V * 2
    # HBM-side bytes per launch: separate rocprofv3 --pmc pass (committed record), quoted for the step's 256 rows
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r1_pmc_epilogue_traffic.json")
    if rows == 256 and os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            traffic = json.load(fh).get("traffic_bytes")
    t_big = timed(2048)
    big_bytes = 2048 * V * 2
    return {"bound": "hbm", "kernel": "remask_stream_kernel<true> + remask_merge_kernel (llada_remask_scores as generate() calls "
                                      "it), low_confidence, temperature 0",
            "achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4),
            "rows_per_launch": rows, "bytes_per_launch": nbytes, "avg_launch_us": round(t * 1e6, 2), "traffic": traffic,
            "large_launch": {"rows_per_launch": 2048, "achieved": round(big_bytes / t_big / 1e9, 1),
                             "frac": round(big_bytes / t_big / 8e12, 4), "avg_launch_us": round(t_big * 1e6, 2)}}


def other_configs(torch, generate, model, cfg, sync_all):
    """The remaining benchmark configurations on this one GPU, a few steps each (the headline `value` stays the
    8-rows-per-GPU line): the 64-prompt batch on ONE GPU at the headline shape and at gen512/blk64 (the N = 1
    points of the strong-scaling runs), guidance + Gumbel noise at gen1024/blk128, and the likelihood."""
    from llada_amd import get_log_likelihood
    out = {}
    gen = torch.Generator().manual_seed(1234)
    prompts = torch.randint(0, 126000, (64, 128), generator=gen, dtype=torch.long)

    def mfma_util(B, L, rows_head, ms):
        d, F, V, nl = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size, cfg.n_layers
        lin = nl * (4 * d * d + 3 * d * F) - (d * d + 3 * d * F)
        fl = 2.0 * B * L * lin + 2.0 * rows_head * (d * d + 3 * d * F) + 2.0 * rows_head * d * V + B * L * 4.0 * L * d * nl
        return round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
    for name, G, bl in (("batch64_gen256_blk32_one_gpu", 256, 32), ("batch64_gen512_blk64_one_gpu", 512, 64)):
        t = timed_steps(generate, model, prompts, G, bl, 2, 6, sync_all)
        ms = 1e3 * t / 6
        out[name] = {"ms_per_step": round(ms, 3), "tokens_per_s": round(64 / (ms * 1e-3), 2),
                     "mfma_util_whole_step": mfma_util(64, 128 + G, 64 * bl, ms),
                     "workload": f"64 prompts x P=128 on ONE GPU, gen_length={G}, steps={G}, block_length={bl}, temperature 0"}
    G, bl = 1024, 128
    am = torch.ones(1, 128, dtype=torch.long)
    t = timed_steps(generate, model, prompts[:1], G, bl, 2, 8, sync_all, attention_mask=am, temperature=0.7, cfg_scale=1.0, seed=99)
    ms = 1e3 * t / 8
    out["cfg1_T0.7_gen1024_blk128"] = {"ms_per_step": round(ms, 3), "tokens_per_s": round(1 / (ms * 1e-3), 2),
                                       "mfma_util_whole_step": mfma_util(2, 128 + G, 2 * bl, ms),
                                       "workload": "1 prompt x P=128, cfg_scale=1.0 (2 x 1152 tokens per forward), temperature 0.7 "
                                                   "(library Philox noise), gen_length=1024, steps=1024, block_length=128"}
    g2 = torch.Generator().manual_seed(4321)
    p_ids = torch.randint(0, 126000, (256,), generator=g2, dtype=torch.long)
    a_ids = torch.randint(0, 126000, (128,), generator=g2, dtype=torch.long)
    get_log_likelihood(model, p_ids, a_ids, mc_num=128, batch_size=32, seed=7)
    sync_all()
    t0 = time.perf_counter()
    val = get_log_likelihood(model, p_ids, a_ids, mc_num=128, batch_size=32, seed=7)
    sync_all()
    out["likelihood_mc128_bs32"] = {"ms_per_call": round(1e3 * (time.perf_counter() - t0), 2), "log_likelihood": round(val, 4),
                                    "workload": "get_log_likelihood(): prompt 256 + answer 128 tokens, mc_num=128, batch_size=32, one GPU"}
    # one rank's share of the same call split over 8 GPUs (4 of the 32 rows of each of the 4 iterations), run alone on
    # this GPU: one fused forward of 16 rows against one forward of 4 rows per iteration
    share = {}
    for tag, fuse in (("fused_forward", True), ("forward_per_iteration", False)):
        kw = dict(mc_num=128, batch_size=32, seed=7, rank=3, world_size=8, return_loss_sums=True, fuse_iterations=fuse)
        sums = get_log_likelihood(model, p_ids, a_ids, **kw)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(3):
            sums = get_log_likelihood(model, p_ids, a_ids, **kw)
        sync_all()
        share[tag] = {"ms_per_call": round(1e3 * (time.perf_counter() - t0) / 3, 2), "loss_sums": [round(float(v), 3) for v in sums.cpu()]}
    # the evaluation harness' generate_until (eval_llada.py:246-279) on 8 requests of different prompt lengths: one request
    # at a time, as the reference runs them, against one left-padded batch with masked padding and per-row positions
    from llada_amd.eval_harness import LLaDAEvalHarness

    class _IdTokenizer:
        """Tokens are written as decimal ids separated by blanks (a benchmark needs no vocabulary)."""

        def __call__(self, text):
            return {"input_ids": [int(t) for t in text.split()]}

        def decode(self, ids, skip_special_tokens=False):
            return " ".join(str(int(i)) for i in ids if not (skip_special_tokens and int(i) >= 126000))

    class _Req:
        def __init__(self, *args):
            self.args = args
    lens = [128, 96, 120, 64, 100, 128, 80, 112]
    reqs = [_Req(" ".join(str(int(v)) for v in prompts[i, :n]), {"until": ["<|never|>"]}) for i, n in enumerate(lens)]
    hk = dict(model=model, tokenizer=_IdTokenizer(), steps=64, gen_length=64, block_length=32, mc_num=1, batch_size=1)
    hres = {}
    for tag, nb in (("one_request_at_a_time", 1), ("batched_8", 8)):
        h = LLaDAEvalHarness(generate_batch_size=nb, **hk)
        h.generate_until(reqs[:2])
        sync_all()
        t0 = time.perf_counter()
        texts = h.generate_until(reqs)
        sync_all()
        dt = time.perf_counter() - t0
        hres[tag] = {"requests_per_s": round(len(reqs) / dt, 3), "tokens_per_s": round(len(reqs) * 64 / dt, 1), "seconds": round(dt, 3)}
        assert len(texts) == len(reqs)
    hres["workload"] = "LLaDAEvalHarness.generate_until: 8 requests, prompts of 64-128 tokens, gen_length=64, steps=64, block_length=32"
    out["harness_generate_until"] = hres
    share["workload"] = "rank 3 of 8 of the call above (4 rows x 4 iterations x 384 tokens), alone on one GPU"
    out["likelihood_mc128_bs32_one_rank_of_8"] = share
    return out


def likelihood_bench(args, torch, dist, model, cfg, world, rank, device, backend):
    """The likelihood benchmark configuration: get_log_likelihood(mc_num=128, batch_size=32) on synthetic ids, the 32 rows of
    every Monte Carlo iteration dealt over the ranks, one all-reduce of the per-iteration loss sums per call."""
    from llada_amd import get_log_likelihood
    gen = torch.Generator().manual_seed(4321)
    prompt = torch.randint(0, 126000, (256,), generator=gen, dtype=torch.long)
    answer = torch.randint(0, 126000, (128,), generator=gen, dtype=torch.long)
    mc_num, batch_size = 128, 32

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def call():
        return get_log_likelihood(model, prompt, answer, mc_num=mc_num, batch_size=batch_size, seed=7, rank=rank,
                                  world_size=world)
    W_, K_ = max(args.warmup, 1), args.steps
    for _ in range(min(W_, 2)):
        val = call()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(K_):
        val = call()
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        l = 384
        flops = (mc_num * l) * 2.0 * (cfg.n_layers * (4 * cfg.d_model ** 2 + 3 * cfg.d_model * cfg.mlp_hidden)) \
            + mc_num * 4.0 * l * l * cfg.d_model * cfg.n_layers
        rec = {"metric": "get_log_likelihood calls/s, LLaDA-8B mc_num=128 batch_size=32 (prompt 256 + answer 128 tokens)",
               "value": round(K_ / elapsed, 4), "unit": "calls/s", "n_gpus": world, "steps": K_, "warmup": min(W_, 2),
               "ms_per_step": round(1e3 * elapsed / K_, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "get_log_likelihood(): mc_num=128, batch_size=32 (4 iterations), 384 tokens per row, rows of "
                                      "every iteration split over the ranks, all-reduce of the loss sums",
                          "layers": cfg.n_layers, "parallelism": f"dp{world}"},
               "rccl_ranks": (dist.get_world_size() if world > 1 else 1) if backend == "nccl" else None,
               "comm_backend": backend if world > 1 else None,
               "log_likelihood": round(val, 4),
               "mfma_util_layers": round(flops / (elapsed / K_) / 1e12 / PEAK_BF16_TFLOPS / world, 4)}
        print(json.dumps(rec), flush=True)


def launch_ranks(n):
    """Parent of a self-launched multi-GPU run: starts one child per rank (the same command line, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set) and waits for them.  This process never touches the GPU (it does not even import
    torch).  Rank 0 inherits stdout, so its JSON line is this command's output.  Returns the exit status: non-zero
    as soon as any rank fails (the others are then stopped)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    status = 0
    live = list(procs)
    while live and status == 0:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is not None:
                live.remove(p)
                if rc != 0:
                    status = rc if rc > 0 else 1
    for p in live:                      # a rank failed: stop the ranks this process started (by handle)
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8, help="prompt rows per GPU (weak scaling, the default)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling: this many prompts in all, dealt over the ranks (overrides --batch)")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--gen", type=int, default=256)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers makes the result INVALID")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--no-whole-call", action="store_true",
                    help="skip the whole-call cross-check of `value` (one full generate() of the headline configuration)")
    ap.add_argument("--workload", choices=["generate", "likelihood"], default="generate",
                    help="likelihood: get_log_likelihood(mc_num=128, batch_size=32), prompt 256 + answer 128 tokens, the MC rows of "
                         "every iteration split over the ranks and the loss sums all-reduced (strong scaling; a step = one call)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))        # before anything here has touched the GPU

    import torch
    import torch.distributed as dist

    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, _lib, generate

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LLADA_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks then share devices and the timing reduce runs on the host); the real run is RCCL
    backend = os.environ.get("LLADA_BENCH_BACKEND", "nccl")
    if os.environ.get("LLADA_BENCH_ECHO_RANK"):
        print(f"bench rank {rank} of {world}", file=sys.stderr, flush=True)
    n_dev = torch.cuda.device_count()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if backend == "nccl" and local_rank >= n_dev:
        sys.exit(f"bench.py: rank {local_rank} has no GPU ({n_dev} visible); LLADA_BENCH_BACKEND=gloo lets ranks share devices")
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    # ranks of the collective backend actually in use (RCCL unless a rehearsal asked for gloo)
    comm_ranks = dist.get_world_size() if world > 1 else 1

    cfg = LLaDAConfig(n_layers=args.layers)
    model = LLaDAModel(LLaDAWeights.random(cfg, device, seed=0))
    lib = _lib.load()

    if args.workload == "likelihood":
        likelihood_bench(args, torch, dist, model, cfg, world, rank, device, backend)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    P, G, bl = args.prompt, args.gen, args.block
    L = P + G
    steps_total = G                                    # steps == gen_length (best-quality setting)
    gen = torch.Generator().manual_seed(1234)
    strong = args.total_batch > 0
    if strong:
        # a fixed set of prompts dealt over the ranks in contiguous, balanced shards (llada_amd.parallel)
        from llada_amd.parallel import shard_bounds
        total_rows = args.total_batch
        if total_rows < world:
            sys.exit(f"bench.py: --total-batch {total_rows} is fewer than one prompt per rank")
        bounds = shard_bounds(total_rows, world)
        row_lo, row_hi = bounds[rank], bounds[rank + 1]
    else:
        total_rows = world * args.batch
        row_lo, row_hi = rank * args.batch, (rank + 1) * args.batch
    B = row_hi - row_lo                                # this rank's rows
    prompts_all = torch.randint(0, 126000, (total_rows, P), generator=gen, dtype=torch.long)
    prompt = prompts_all[row_lo:row_hi]

    W_, K_ = args.warmup, args.steps

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    elapsed = timed_steps(generate, model, prompt, G, bl, W_, K_, sync_all,
                          on_start=lambda: lib.llada_gemm_profile_begin(_lib.EPI_SWIGLU, K_ * cfg.n_layers, B * L),
                          row_offset=row_lo, span_block=True)
    # cross-check of the extrapolation value = rows * G / (steps_total * t_step): one WHOLE generate() call of the
    # headline configuration, wall clock between barrier + synchronize pairs (every block boundary, allocation
    # and host read of the call included)
    def run_whole_call():
        sync_all()
        t0 = time.perf_counter()
        generate(model, prompt, None, steps=steps_total, gen_length=G, block_length=bl, temperature=0., cfg_scale=0.,
                 remasking="low_confidence", row_offset=row_lo)
        sync_all()
        return time.perf_counter() - t0

    whole_call_s, whole_call_attempts = None, 0
    under_load = {}
    if not args.no_whole_call:
        sampler_thread = None
        if rank == 0:      # clock and power a few seconds into the whole call (host-side tools, nothing on the GPU)
            import threading
            sampler_thread = threading.Thread(target=sample_clock_power, args=(3.0, under_load), daemon=True)
            sampler_thread.start()
        whole_call_s, whole_call_attempts = run_whole_call(), 1
        if sampler_thread is not None:
            sampler_thread.join(timeout=30)
    tot_ms, n_s = C.c_double(0), C.c_int(0)
    lib.llada_gemm_profile_end(C.byref(tot_ms), C.byref(n_s))
    torch.cuda.synchronize()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / K_
    tokens_per_s = total_rows * G / (steps_total * ms_per_step * 1e-3)
    if whole_call_s is not None and world == 1 and abs(total_rows * G / whole_call_s / tokens_per_s - 1.0) > 0.03:
        # one GPU: a single disagreement may be a transient of the box (8 s of sustained load against a 2 s window);
        # the call is timed once more before the run is failed
        whole_call_s, whole_call_attempts = run_whole_call(), 2
    if whole_call_s is not None:
        tw = torch.tensor([whole_call_s], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        whole_call_s = float(tw.item())

    if rank == 0:
        M = B * L
        F, d, V = cfg.mlp_hidden, cfg.d_model, cfg.vocab_size
        gemm_flops = 2.0 * M * (2 * F) * d
        gemm_ms = tot_ms.value / max(n_s.value, 1)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # executed FLOPs of one step: all rows through every layer, except that the last layer's
        # output projection and MLP (and the LM head) only see the B*bl block rows
        R = B * bl
        lin = cfg.n_layers * (4 * d * d + 3 * d * F) - (d * d + 3 * d * F)
        flops_step = 2.0 * M * lin + 2.0 * R * (d * d + 3 * d * F) + 2.0 * R * d * V + M * 4.0 * L * d * cfg.n_layers
        # HBM-side bytes per launch of that kernel come from separate rocprofv3 --pmc passes (counters
        # cannot be read from inside the process); the committed record is quoted, null without it
        traffic, traffic_source = None, None
        pmc_name = "r3_pmc_gateup_traffic.json"
        pmc_file = os.path.join(ROOT, "profiles", pmc_name)
        if B * L == 3072 and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            traffic = pmc.get("traffic_bytes")
            traffic_source = {"file": "profiles/" + pmc_name, "date": pmc.get("date"), "command": pmc.get("command"),
                              "read_bytes": pmc.get("read_bytes"), "write_bytes": pmc.get("write_bytes"),
                              "algorithmic_bytes": pmc.get("algorithmic_bytes"),
                              "note": "separate rocprofv3 --pmc passes on the same kernel and shape (counters cannot be read from "
                                      "inside the timed process); memory-side requests of the L2s, Infinity-Cache hits included"}
        rec = {
            "metric": f"generated tokens/sec/node, LLaDA-8B gen{G} steps{steps_total} blk{bl}; ms/denoise step",
            "value": round(tokens_per_s, 3), "unit": "tokens/s", "n_gpus": world, "steps": K_, "warmup": W_,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "timed_window": {"first_step": window_start(W_, K_, bl, G), "steps": K_, "steps_per_block": bl,
                             "spans_block_boundary": window_start(W_, K_, bl, G) // bl != (window_start(W_, K_, bl, G) + K_ - 1) // bl},
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"LLaDA-8B generate(): {total_rows} prompts dealt over {world} GPU(s) ({B} on rank 0) x P={P}, "
                                    if strong else f"LLaDA-8B generate(): {B} prompts/GPU x P={P}, ")
                                   + f"gen_length={G}, steps={steps_total}, block_length={bl}, temperature 0, low_confidence"
                                   + ("" if strong else " (north-star batch 64 = 8 GPUs x 8)"),
                       "layers": cfg.n_layers, "rows_per_gpu": B, "total_rows": total_rows, "seq_len": L,
                       "parallelism": f"dp{world}"},
            "rccl_ranks": comm_ranks if backend == "nccl" else None, "comm_backend": backend if world > 1 else None,
            "step_tflops_executed": round(flops_step / 1e12, 3),
            "mfma_util_whole_step": round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_stream_kernel<6,SWIGLU> (gate/up projection, M = rows_per_gpu x seq_len)",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "launches_timed": n_s.value, "avg_launch_ms": round(gemm_ms, 4),
                         "flops_per_launch": gemm_flops},
        }
        if whole_call_s is not None:
            v_whole = total_rows * G / whole_call_s
            rec["whole_call_s"] = round(whole_call_s, 4)
            rec["whole_call_attempts"] = whole_call_attempts
            if under_load:
                rec["under_load"] = dict(under_load, note="rocm-smi / amd-smi 3 s into the whole call; the bf16 peak of "
                                         "`roofline` is quoted at 2.4 GHz")
                if under_load.get("sclk_mhz") and rec["roofline"].get("achieved"):
                    peak_at = rec["roofline"]["peak"] * under_load["sclk_mhz"] / 2400.0
                    rec["under_load"]["mfma_peak_at_sclk_tflops"] = round(peak_at, 1)
                    rec["under_load"]["roofline_frac_at_sclk"] = round(rec["roofline"]["achieved"] / peak_at, 4)
            rec["value_from_whole_call"] = round(v_whole, 3)
            rec["whole_call_vs_value"] = round(v_whole / tokens_per_s, 4)
            rec["whole_call_check"] = "ok" if abs(v_whole / tokens_per_s - 1.0) <= 0.03 else "FAILED: differs from value by more than 3 %"
            if rec["whole_call_check"] != "ok":
                msg = (f"bench.py: value extrapolated from {K_} timed steps ({tokens_per_s:.2f} tokens/s) and the whole "
                       f"{steps_total}-step call ({v_whole:.2f} tokens/s) differ by more than 3 %")
                if world == 1:
                    sys.exit(msg + ": " + json.dumps(rec))        # one GPU: nothing but this process moves the clock
                # several GPUs under load share a node's power budget: the line is still printed, flagged
                print(msg, file=sys.stderr, flush=True)
        rec["measured_ceilings"] = measured_ceilings(torch, device, M, 2 * F, d)
        rec["epilogue_roofline"] = epilogue_roofline(torch, lib, _lib, device, R, V, cfg.mask_token_id)
        if world == 1 and B != 1 and not args.no_latency:
            # the single-sequence configuration (batch 1, same gen_length / steps / block_length):
            # per-step latency of forward + epilogue
            t1_sync = timed_steps(generate, model, prompt[:1], G, bl, W_, K_, sync_all, span_block=True)
            t1 = event_timed_steps(torch, generate, model, prompt[:1], G, bl, max(W_, 8), K_)
            s1 = t1 / K_
            # one sequence sits on the ridge: report it against both bounds.  Bytes = every weight matrix read once
            # (the LM head and the last layer's three late projections serve only the block's rows, but their
            # weights are still read whole); FLOPs = those actually executed.
            w_bytes = 2.0 * cfg.linear_params()
            fl1 = 2.0 * L * lin + 2.0 * bl * (d * d + 3 * d * F) + 2.0 * bl * d * V + L * 4.0 * L * d * cfg.n_layers
            rec["latency_batch1"] = {"ms_per_step": round(1e3 * s1, 4), "tokens_per_s": round(G / (steps_total * s1), 2),
                                     "timing": f"HIP events around {K_} steps inside one uninterrupted generate() call (window over a "
                                               "block boundary); ms_per_step_sync_bracketed is the same window between host "
                                               "synchronisations, which charges the short steps the device's restart after idling",
                                     "ms_per_step_sync_bracketed": round(1e3 * t1_sync / K_, 4),
                                     "workload": f"1 prompt x P={P}, gen_length={G}, steps={steps_total}, block_length={bl}",
                                     "roofline": {"weights_GBps": round(w_bytes / s1 / 1e9, 1), "frac_hbm": round(w_bytes / s1 / 8e12, 4),
                                                  "tflops_executed": round(fl1 / 1e12, 3),
                                                  "frac_mfma": round(fl1 / s1 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                  "weight_bytes_per_step": w_bytes}}
        if world == 1 and not args.no_latency:
            # opt-in approximation outside the reference (cache="dual"): whole generate() calls, wall clock
            def whole_call(p_, kind="dual"):
                generate(model, p_, None, steps=G, gen_length=G, block_length=bl, cache=kind)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                generate(model, p_, None, steps=G, gen_length=G, block_length=bl, cache=kind)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            tc = whole_call(prompt)
            rec["block_cached"] = {"note": "generate(cache='dual' | 'prefix'): APPROXIMATIONS, not the reference's algorithm; never `value`",
                                   "tokens_per_s": round(B * G / tc, 1), "ms_per_step_mean": round(1e3 * tc / steps_total, 3),
                                   "seconds_per_call": round(tc, 3)}
            tp = whole_call(prompt, "prefix")
            rec["block_cached"]["prefix_tokens_per_s"] = round(B * G / tp, 1)
            rec["block_cached"]["prefix_ms_per_step_mean"] = round(1e3 * tp / steps_total, 3)
            if B != 1:
                t1c = whole_call(prompt[:1])
                rec["block_cached"]["batch1_tokens_per_s"] = round(G / t1c, 1)
                rec["block_cached"]["batch1_ms_per_step_mean"] = round(1e3 * t1c / steps_total, 3)
                rec["block_cached"]["batch1_prefix_tokens_per_s"] = round(G / whole_call(prompt[:1], "prefix"), 1)
        if world == 1 and not strong and B == 8 and (P, G, bl) == (128, 256, 32) and not args.no_latency:
            rec["other_configs"] = other_configs(torch, generate, model, cfg, sync_all)
        if world == 1 and not args.no_cpu_baseline:      # the host baseline is reported at N = 1 only
            rec["cpu_baseline"] = cpu_baseline(cfg)
            # the same configuration on the HIP path (batch 1, P = 128, gen_length 64,
            # steps 64, block_length 32): a whole call, wall clock.  `value` above is another workload (8 rows per
            # GPU, gen_length 256); this is the like-for-like pair of the host number.
            p1 = prompt[:1, :128] if P >= 128 else prompt[:1]
            generate(model, p1, None, steps=64, gen_length=64, block_length=32, temperature=0., remasking="low_confidence")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            generate(model, p1, None, steps=64, gen_length=64, block_length=32, temperature=0., remasking="low_confidence")
            torch.cuda.synchronize()
            tg = time.perf_counter() - t0
            rec["cpu_baseline"]["comparable_to"] = "cpu_baseline.gpu_same_config (NOT `value`, which is 8 rows per GPU at gen_length 256)"
            rec["cpu_baseline"]["gpu_same_config"] = {"tokens_per_s": round(64 / tg, 2), "ms_per_step": round(1e3 * tg / 64, 3),
                                                      "seconds_per_call": round(tg, 4),
                                                      "workload": "batch 1, P=128, gen_length=64, steps=64, block_length=32, T=0, "
                                                                  "low_confidence: one whole generate() call on the HIP path"}
            rec["cpu_baseline"]["speedup_same_config"] = round((64 / tg) / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
