"""ms per denoising step of generate() at a given shape, for A/B runs of two builds on one box
(LLADA_HIP_LIB=<other .so> selects the build): python tools/step_time.py B [P G block steps]"""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, generate

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P, G, bl, K = (int(v) for v in (sys.argv[2:6] + ["128", "256", "32", "48"][len(sys.argv[2:6]):]))
torch.cuda.set_device(0)
model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda", seed=0))
prompt = torch.randint(0, 126000, (B, P), generator=torch.Generator().manual_seed(1234))
for rnd in range(3):
    t = bench.timed_steps(generate, model, prompt, G, bl, 8, K, torch.cuda.synchronize)
    print(f"B={B} P={P} G={G} block={bl}: {1e3 * t / K:.4f} ms/step", flush=True)
