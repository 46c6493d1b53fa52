"""In-situ A/B of a library switch: time the full 32-layer forward (8 x 384 tokens, block-row LM head)
alternating hook(v0) / hook(v1) in one process.  usage: forward_ab.py <hook name> [rounds] [rows] [v0] [v1]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, _lib
lib = _lib.load()
hook = getattr(lib, sys.argv[1]); hook.argtypes = [C.c_int]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
V0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
V1 = int(sys.argv[5]) if len(sys.argv) > 5 else 1
model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda"))
L = 384
ids = torch.randint(0, 126000, (B, L), device="cuda")
rows = (torch.arange(B, device="cuda")[:, None] * L + torch.arange(128, 160, device="cuda")[None]).reshape(-1).int()
out = None
res = {V0: [], V1: []}
for r in range(rounds):
    for v in (V0, V1):
        hook(v)
        for _ in range(2): out = model.forward_rows(ids, None, rows, out=out)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): out = model.forward_rows(ids, None, rows, out=out)
        e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 8)
hook(V0)
for v in (V0, V1): print(f"{sys.argv[1]}({v}): median {np.median(res[v]):.3f} ms  min {np.min(res[v]):.3f} ms  all {[round(x, 2) for x in res[v]]}")
