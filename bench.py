#!/usr/bin/env python3
"""Headline benchmark: LLaDA-8B generate(), gen_length=256, steps=256, block_length=32, synthetic
prompts (P=128) and random-init weights at the true shapes, data-parallel over N GPUs with a fixed
number of prompt rows per GPU (weak scaling; 8 rows/GPU = the 64-prompt north-star batch on 8 GPUs).

One "step" = one denoising step of the whole per-GPU batch: the non-causal forward over all
B*(P+G) tokens plus the fused remasking epilogue.  `value` = generated tokens/s of the whole job =
N*B*G / (steps_total * time_per_step).

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


def cpu_baseline(B, L, bl, cfg, seconds_hint=20.0):
    """Oracle (numpy) on the host: TWO transformer layers of the 32 at the full token count, plus the
    final norm / LM head on the block rows and the remasking epilogue; the step time is
    extrapolated as 32 * layer + head + epilogue.  kind = "port" (the oracle restates the
    reference's algorithm; the reference's own model code is not in its tree)."""
    import numpy as np

    from oracle import sampler as osamp
    from oracle.model import LLaDAConfig as OCfg
    from oracle.model import forward, rmsnorm
    from oracle.bf16 import round_bf16
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    n_sample = 2                                    # layers actually run (about 11 s on 16 host threads)
    ocfg = OCfg(d_model=cfg.d_model, n_layers=n_sample, n_heads=cfg.n_heads, mlp_hidden=cfg.mlp_hidden, vocab_size=cfg.vocab_size)
    g = np.random.Generator(np.random.PCG64(0))

    def w(n, k, s=0.02):
        return round_bf16(g.standard_normal(size=(n, k), dtype=np.float32) * np.float32(s))

    d, F, V = cfg.d_model, cfg.mlp_hidden, cfg.vocab_size
    # a small embedding table (ids are folded into it) keeps the sample's memory bounded
    W = {"wte": w(4096, d), "ln_f": np.ones(d, np.float32), "ff_out": w(V, d, 0.16)}
    for i in range(n_sample):
        p = f"blocks.{i}."
        for name, shape in (("q_proj", (d, d)), ("k_proj", (d, d)), ("v_proj", (d, d)), ("attn_out", (d, d)),
                            ("ff_proj", (F, d)), ("up_proj", (F, d)), ("ff_out", (d, F))):
            W[p + name] = w(*shape)
        W[p + "attn_norm"] = np.ones(d, np.float32)
        W[p + "ff_norm"] = np.ones(d, np.float32)
    ids = g.integers(0, 4096, size=(B, L), dtype=np.int64)
    t0 = time.perf_counter()
    h = forward(W, ocfg, ids, None, emulate_bf16=True, return_hidden=True)     # n_sample layers + final norm
    t_layer = (time.perf_counter() - t0) / n_sample
    t0 = time.perf_counter()
    hb = h[:, L - bl:, :].reshape(B * bl, d)
    logits = round_bf16(hb @ W["ff_out"].T).reshape(B, bl, V)
    x = np.full((B, bl), cfg.mask_token_id, dtype=np.int64)
    x0, conf = osamp.confidence_step(x, logits, cfg.mask_token_id, bl, "low_confidence")
    osamp.topk_transfer(x, x0, conf, np.ones(B, dtype=np.int64))
    t_head = time.perf_counter() - t0
    t_step = cfg.n_layers * t_layer + t_head
    return t_step, cores, (f"{n_sample} of {cfg.n_layers} layers at {B}x{L} tokens ({t_layer:.2f} s each) + LM head/epilogue on "
                           f"{B * bl} rows ({t_head:.2f} s); step = {cfg.n_layers}*layer + head")


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


def timed_steps(generate, model, prompt, G, bl, W_, K_, sync_all, on_start=None, row_offset=0):
    """Runs generate() (steps == gen_length) and times steps W_ .. W_+K_ of it between two
    barrier + synchronize pairs; returns seconds for the K_ steps."""
    state = {"n": 0, "t0": None, "t1": None}

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

    if W_ == 0:
        begin()
    while state["t1"] is None:
        generate(model, prompt, None, steps=G, gen_length=G, block_length=bl, temperature=0., cfg_scale=0.,
                 remasking="low_confidence", row_offset=row_offset, step_callback=cb)
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


def likelihood_bench(args, torch, dist, model, cfg, world, rank, device, backend):
    """BASELINE configuration 4: get_log_likelihood(mc_num=128, batch_size=32) on synthetic ids, the 32 rows of
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
               "log_likelihood": round(val, 4),
               "mfma_util_layers": round(flops / (elapsed / K_) / 1e12 / PEAK_BF16_TFLOPS / world, 4)}
        print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8, help="prompt rows per GPU")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--gen", type=int, default=256)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers makes the result INVALID")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--workload", choices=["generate", "likelihood"], default="generate",
                    help="likelihood: get_log_likelihood(mc_num=128, batch_size=32), prompt 256 + answer 128 tokens, the MC rows of "
                         "every iteration split over the ranks and the loss sums all-reduced (strong scaling; a step = one call)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, _lib, generate

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LLADA_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks then share devices and the timing reduce runs on the host); the real run is RCCL
    backend = os.environ.get("LLADA_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    cfg = LLaDAConfig(n_layers=args.layers)
    model = LLaDAModel(LLaDAWeights.random(cfg, device, seed=0))
    lib = _lib.load()

    if args.workload == "likelihood":
        likelihood_bench(args, torch, dist, model, cfg, world, rank, device, backend)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    B, P, G, bl = args.batch, args.prompt, args.gen, args.block
    L = P + G
    steps_total = G                                    # steps == gen_length (best-quality setting)
    gen = torch.Generator().manual_seed(1234)
    prompts_all = torch.randint(0, 126000, (world * B, P), generator=gen, dtype=torch.long)
    prompt = prompts_all[rank * B:(rank + 1) * B]

    W_, K_ = args.warmup, args.steps

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    elapsed = timed_steps(generate, model, prompt, G, bl, W_, K_, sync_all,
                          on_start=lambda: lib.llada_gemm_profile_begin(_lib.EPI_SWIGLU, K_ * cfg.n_layers, B * L),
                          row_offset=rank * B)
    tot_ms, n_s = C.c_double(0), C.c_int(0)
    lib.llada_gemm_profile_end(C.byref(tot_ms), C.byref(n_s))
    torch.cuda.synchronize()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / K_
    tokens_per_s = world * B * G / (steps_total * ms_per_step * 1e-3)

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
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r1_pmc_gateup_traffic.json")
        if B * L == 3072 and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                traffic = json.load(fh).get("traffic_bytes")
        rec = {
            "metric": "generated tokens/sec/node, LLaDA-8B gen256 steps256 blk32; ms/denoise step",
            "value": round(tokens_per_s, 3), "unit": "tokens/s", "n_gpus": world, "steps": K_, "warmup": W_,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"LLaDA-8B generate(): {B} prompts/GPU x P={P}, gen_length={G}, steps={steps_total}, "
                                   f"block_length={bl}, temperature 0, low_confidence (north-star batch 64 = 8 GPUs x 8)",
                       "layers": cfg.n_layers, "rows_per_gpu": B, "seq_len": L, "parallelism": f"dp{world}"},
            "step_tflops_executed": round(flops_step / 1e12, 3),
            "mfma_util_whole_step": round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_stream_kernel<6,SWIGLU> (gate/up projection, M = rows_per_gpu x seq_len)",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "launches_timed": n_s.value, "avg_launch_ms": round(gemm_ms, 4),
                         "flops_per_launch": gemm_flops},
        }
        rec["measured_ceilings"] = measured_ceilings(torch, device, M, 2 * F, d)
        rec["epilogue_roofline"] = epilogue_roofline(torch, lib, _lib, device, R, V, cfg.mask_token_id)
        if world == 1 and B != 1 and not args.no_latency:
            # the single-sequence configuration (batch 1, same gen_length / steps / block_length):
            # per-step latency of forward + epilogue
            t1 = timed_steps(generate, model, prompt[:1], G, bl, W_, K_, sync_all)
            rec["latency_batch1"] = {"ms_per_step": round(1e3 * t1 / K_, 4), "tokens_per_s": round(G / (steps_total * t1 / K_), 2),
                                     "workload": f"1 prompt x P={P}, gen_length={G}, steps={steps_total}, block_length={bl}"}
        if world == 1 and not args.no_latency:
            # opt-in approximation outside the reference (cache="dual"): whole generate() calls, wall clock
            def whole_call(p_):
                generate(model, p_, None, steps=G, gen_length=G, block_length=bl, cache="dual")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                generate(model, p_, None, steps=G, gen_length=G, block_length=bl, cache="dual")
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            tc = whole_call(prompt)
            rec["block_cached"] = {"note": "generate(cache='dual'): APPROXIMATION, not the reference's algorithm; never `value`",
                                   "tokens_per_s": round(B * G / tc, 1), "ms_per_step_mean": round(1e3 * tc / steps_total, 3),
                                   "seconds_per_call": round(tc, 3)}
            if B != 1:
                t1c = whole_call(prompt[:1])
                rec["block_cached"]["batch1_tokens_per_s"] = round(G / t1c, 1)
                rec["block_cached"]["batch1_ms_per_step_mean"] = round(1e3 * t1c / steps_total, 3)
        if world == 1 and not args.no_cpu_baseline:      # the host baseline is reported at N = 1 only
            t_step, cores, sample = cpu_baseline(B, L, bl, cfg)
            rec["cpu_baseline"] = {"value": round(B * G / (steps_total * t_step), 4), "unit": "tokens/s", "cores": cores,
                                   "kind": "port", "sample": sample, "ms_per_step": round(1e3 * t_step, 1)}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
