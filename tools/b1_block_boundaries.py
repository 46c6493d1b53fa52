import sys, time
sys.path.insert(0, ".")
import torch
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, generate
torch.cuda.set_device(0)
model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda", seed=0))
prompt = torch.randint(0, 126000, (1, 128), generator=torch.Generator().manual_seed(1234))
for rnd in range(3):
    ev = []
    def cb(b, s, x, x0, c):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((b, s, e, time.perf_counter()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    generate(model, prompt, steps=256, gen_length=256, block_length=32, step_callback=cb)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    gaps = [ev[i][2].elapsed_time(ev[i + 1][2]) for i in range(len(ev) - 1)]
    inner = [g for i, g in enumerate(gaps) if ev[i + 1][1] != 0]
    bound = [g for i, g in enumerate(gaps) if ev[i + 1][1] == 0]
    cpu = [(ev[i + 1][3] - ev[i][3]) * 1e3 for i in range(len(ev) - 1)]
    print(f"whole call {t1*1e3:.1f} ms = {t1*1e3/256:.3f} ms/step; enqueue finished after {t_enq*1e3:.1f} ms; "
          f"GPU step inside a block {sum(inner)/len(inner):.3f} ms, first step of a block {sum(bound)/len(bound):.3f} ms; "
          f"host per step {sum(cpu)/len(cpu):.3f} ms (max {max(cpu):.2f})", flush=True)
