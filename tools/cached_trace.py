"""One generate(cache='dual') call at the benchmark shapes (for rocprofv3 --kernel-trace): cached_trace.py [batch]."""
import sys, time, torch
sys.path.insert(0, ".")
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, generate
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = LLaDAConfig()
model = LLaDAModel(LLaDAWeights.random(cfg, "cuda", seed=0))
prompt = torch.randint(0, 126000, (B, 128), generator=torch.Generator().manual_seed(1234), dtype=torch.long)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    generate(model, prompt, None, steps=256, gen_length=256, block_length=32, cache="dual")
    torch.cuda.synchronize(); print(f"B={B}: {time.perf_counter() - t0:.3f} s per call", flush=True)
