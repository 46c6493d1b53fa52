"""The four projection shapes at M = 3072 with and without the write-out (test hook llada_gemm_no_store): cost of the epilogue stores."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); lib.llada_gemm_no_store.argtypes = [C.c_int]
M = 3072
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (24576, 4096), "down": (4096, 12288)}.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out = []
    for sk in (0,):
        for ns in (0, 1):
            lib.llada_gemm_no_store(ns); ts = []
            for r in range(5):
                for _ in range(3): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 20
                e0.record()
                for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
            out.append(f"sk{sk}/nostore{ns}: {np.median(ts):7.1f}us")
    print(name, "  ".join(out), flush=True)
lib.llada_gemm_no_store(0)
