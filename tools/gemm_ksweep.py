"""GEMM time against K at M = 3072 (pipelined and simple loop): per-K-step time shows the fixed fill / drain / write-out cost.  gemm_ksweep.py [N]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); pass
M = 3072; N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for pipe in (1, 0):
    lib.llada_gemm_use_pipelined(pipe); lib.llada_gemm_force_tile_rows(192)
    for K in (256, 512, 1024, 2048, 4096, 8192, 16384):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ts = []
        for r in range(5):
            for _ in range(3): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 20
            e0.record()
            for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
        t = np.median(ts); print(f"pipe={pipe} N={N} K={K:6d} {t:8.1f} us  {2.0*M*N*K/t/1e6:8.1f} TF  per-kstep {t/(K/64)*1e3:7.1f} ns", flush=True)
