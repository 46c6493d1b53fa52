"""Per-step / per-call timings of the other benchmark configurations on one GPU (random-init LLaDA-8B,
synthetic prompts): long generation, CFG + Gumbel sampling, Monte Carlo likelihood."""
import json, sys, time, torch
sys.path.insert(0, ".")
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, generate, get_log_likelihood

model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda"))
g = torch.Generator().manual_seed(7)


class Done(Exception):
    pass


def time_generate(name, B, P, G, bl, temperature, cfg_scale, warm=4, steps=24, mask=False):
    prompt = torch.randint(0, 126000, (B, P), generator=g)
    st = {"n": 0}

    def cb(block, step, x, x0, conf):
        st["n"] += 1
        if st["n"] == warm:
            torch.cuda.synchronize(); st["t0"] = time.perf_counter()
        elif st["n"] == warm + steps:
            torch.cuda.synchronize(); st["t1"] = time.perf_counter()
            raise Done

    am = torch.ones(B, P, dtype=torch.long) if (mask or cfg_scale > 0) else None
    try:
        generate(model, prompt, am, steps=G, gen_length=G, block_length=bl, temperature=temperature, cfg_scale=cfg_scale,
                 remasking="low_confidence", step_callback=cb)
    except Done:
        pass
    ms = 1e3 * (st["t1"] - st["t0"]) / steps
    print(json.dumps({"config": name, "rows": B, "seq_len": P + G, "block": bl, "temperature": temperature,
                      "cfg_scale": cfg_scale, "ms_per_step": round(ms, 3),
                      "tokens_per_s_full_run": round(B * G / (G * ms * 1e-3), 2)}), flush=True)


time_generate("cfg3 gen512/steps512/blk64, 8 rows/GPU", 8, 128, 512, 64, 0.0, 0.0)
time_generate("cfg5 cfg_scale 1.0 + T=0.7, gen1024/blk128, B=1", 1, 128, 1024, 128, 0.7, 1.0)
time_generate("cfg2 B=1 gen256/blk32 with T=0.7 (Gumbel epilogue)", 1, 128, 256, 32, 0.7, 0.0)

prompt = torch.randint(0, 126000, (128,), generator=g)
answer = torch.randint(0, 126000, (256,), generator=g)
for cfg_scale in (0.0, 0.5):
    get_log_likelihood(model, prompt, answer, mc_num=32, batch_size=32, cfg_scale=cfg_scale)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = get_log_likelihood(model, prompt, answer, mc_num=128, batch_size=32, cfg_scale=cfg_scale)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"config": "cfg4 get_log_likelihood mc_num=128 batch_size=32, prompt 128 + answer 256", "cfg_scale": cfg_scale,
                      "seconds_per_call": round(dt, 4), "mc_samples_per_s": round(128 / dt, 1), "value": round(v, 3)}), flush=True)
