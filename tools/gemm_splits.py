"""Small-M GEMM sweep (weights warm): tile rows x forced split-K count, microseconds per launch against the weight-streaming bound.  gemm_splits.py [M]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (24576, 4096), "down": (4096, 12288)}.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    out = []
    for rows in (128, 192):
        for S in (0, 1, 2, 3, 4, 6, 8, 12, 16):
            if S > 1 and (K // 64) % S: continue
            lib.llada_gemm_force_tile_rows(rows); lib.llada_gemm_force_splits(S if S else 0); ts = []
            wsp, wsb = (ws.data_ptr(), ws.numel() * 4) if S != 1 else (None, 0)
            for r in range(3):
                for _ in range(3): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, wsp, wsb, None)
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 20
                e0.record()
                for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, wsp, wsb, None)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
            out.append(f"r{rows}/S{S if S else 'auto'}:{np.median(ts):6.1f}")
    print(name, f"(W {N*K*2/1e6:.0f} MB, ideal {N*K*2/5e6:.0f}us)", " ".join(out), flush=True)
lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_force_splits(0)
