"""Seconds per whole generate() call with a block cache (or none), for A/B runs of two builds on one box
(LLADA_HIP_LIB=<other .so>): python tools/cached_time.py B [cache=dual|prefix|none] [P G block]"""
import sys, time
sys.path.insert(0, ".")
import torch
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, generate

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cache = sys.argv[2] if len(sys.argv) > 2 else "dual"
P, G, bl = (int(v) for v in (sys.argv[3:6] + ["128", "256", "32"][len(sys.argv[3:6]):]))
torch.cuda.set_device(0)
model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda", seed=0))
prompt = torch.randint(0, 126000, (B, P), generator=torch.Generator().manual_seed(1234))
kw = {} if cache == "none" else {"cache": cache}
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    generate(model, prompt, None, steps=G, gen_length=G, block_length=bl, temperature=0., cfg_scale=0., remasking="low_confidence", **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} cache={cache} P={P} G={G} block={bl}: {dt:.4f} s per call, {1e3 * dt / G:.4f} ms/step, {B * G / dt:.1f} tokens/s", flush=True)
