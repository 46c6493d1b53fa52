"""GEMM micro-benchmark through the C ABI: the four projection shapes of one LLaDA-8B layer at
M = rows, random data, interleaved variants in one process (median of rounds)."""
import ctypes as C
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
pass
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["p192", "s192", "p128", "s256"]
shapes = {"qkv": (12288, 4096, 0), "o": (4096, 4096, 0), "gu": (24576, 4096, 0), "down": (4096, 12288, 0)}
torch.manual_seed(0)
res = {}
for name, (N, K, epi) in shapes.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    flops = 2.0 * M * N * K
    times = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            lib.llada_gemm_use_pipelined(1 if v[0] == "p" else 0)
            lib.llada_gemm_force_tile_rows(int(v[1:]))
            for _ in range(3):
                lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, epi, None, 0, None, None, 0, None, 0, None)
            torch.cuda.synchronize()
            n = 20
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, epi, None, 0, None, None, 0, None, 0, None)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / n)
    print(name, M, N, K, " ".join(f"{v}: {np.median(t)*1e3:7.1f}us {flops/np.median(t)/1e9:7.1f}TF" for v, t in times.items()), flush=True)
lib.llada_gemm_use_pipelined(1); lib.llada_gemm_force_tile_rows(0)
