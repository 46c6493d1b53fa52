"""The four projection GEMMs of a layer (tile-major weights, the model's epilogues, split-K workspace given) at a
mid-size M: the automatic plan against every forced (tile rows, split count), interleaved rounds, median.
gemm_mid_m.py M [rounds]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 1152 if M % 1152 == 0 else 384
shapes = {"qkv": (12288, 4096, 2), "o": (4096, 4096, 1), "gu": (24576, 4096, 3), "down": (4096, 12288, 1)}
torch.manual_seed(0)
cos = torch.rand(L, 64, device="cuda"); sin = torch.rand(L, 64, device="cuda")
for name, (N, K, epi) in shapes.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
    ldc = N // 2 if epi == 3 else N
    Cc = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
    R = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    variants = [(0, 0)] + [(r, s) for r in (128, 192, 256) for s in (0, 2, 4)]
    times = {v: [] for v in variants}

    def call():
        lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, ldc, epi | 0x100, _lib.ptr(R), N if epi == 1 else 0,
                            cos.data_ptr() if epi == 2 else None, sin.data_ptr() if epi == 2 else None, L if epi == 2 else 0,
                            ws.data_ptr(), ws.numel() * 4, None)
    for r in range(rounds):
        for v in variants:
            lib.llada_gemm_force_tile_rows(v[0]); lib.llada_gemm_force_splits(v[1])
            for _ in range(2): call()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_force_splits(0)
    print(f"M={M} {name:5s}", " ".join(f"{'auto' if v == (0, 0) else f'{v[0]}/S{v[1]}'}:{np.median(t):7.1f}" for v, t in times.items()), "us", flush=True)
