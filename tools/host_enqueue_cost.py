"""Host time to enqueue one batch-1 forward (about 230 kernel launches through one C call) against the GPU time it takes:
how far the host runs ahead of the device in the step loop."""
import sys, time
sys.path.insert(0, ".")
import torch
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
torch.cuda.set_device(0)
model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda", seed=0))
for B, L in ((1, 384), (8, 384)):
    x = torch.randint(0, 126000, (B, L), device="cuda")
    rows = (torch.arange(B, dtype=torch.int32, device="cuda")[:, None] * L + torch.arange(128, 160, dtype=torch.int32, device="cuda")[None, :]).reshape(-1).contiguous()
    out = torch.empty(rows.numel(), 126464, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): model.forward_rows(x, None, rows, out=out)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.forward_rows(x, None, rows, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    host.sort(); total.sort()
    print(f"{B} x {L}: host enqueue {host[5]:.3f} ms (min {host[0]:.3f}), forward start to finish {total[5]:.3f} ms", flush=True)
