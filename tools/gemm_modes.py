"""Compare the GEMM main loops (0 simple, 1 quarter-phase pipelined, 2 ring): equality and time."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); lib.llada_gemm_no_store.argtypes = [C.c_int]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rows = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [192, 128]
modes = [int(r) for r in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
EPI = int(sys.argv[4]) if len(sys.argv) > 4 else 0
def call(A, W, Cc, N, K): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N // 2 if EPI == 3 else N, EPI, None, 0, None, None, 0, None, 0, None)
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (24576, 4096), "down": (4096, 12288)}.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ref = None
    for tr in rows:
        lib.llada_gemm_force_tile_rows(tr)
        for mode in modes:
            lib.llada_gemm_use_pipelined(mode)
            Cc = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            call(A, W, Cc, N, K); torch.cuda.synchronize()
            if ref is None: ref = Cc.clone()
            same = torch.equal(ref, Cc)
            res = []
            for ns in (0, 1, 2):
                lib.llada_gemm_no_store(ns); ts = []
                for r in range(4):
                    for _ in range(3): call(A, W, Cc, N, K)
                    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 20
                    e0.record()
                    for _ in range(n): call(A, W, Cc, N, K)
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
                res.append(np.median(ts))
            lib.llada_gemm_no_store(0)
            print(f"{name:5s} rows {tr} mode {mode}: {res[0]:7.1f} us ({2*M*N*K/res[0]*1e-6:6.0f} TF)  no-store {res[1]:7.1f} us  math-only {res[2]:7.1f} us  equal={same}", flush=True)
lib.llada_gemm_use_pipelined(1); lib.llada_gemm_force_tile_rows(0)
