"""Tile-walk sweep at M = 3072 (test hook llada_gemm_tile_walk): M tiles per band, XCD-aware block map on / off; TFLOP/s per shape."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); pass
M = 3072
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (24576, 4096), "down": (4096, 12288)}.items():
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = []
    for (band, remap) in [(8, 1), (8, 0), (4, 1), (16, 1), (2, 1), (16, 0)]:
        lib.llada_gemm_tile_walk(band, remap); ts = []
        for r in range(4):
            for _ in range(3): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 20
            e0.record()
            for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
        out.append(f"band{band}/remap{remap}: {2.0*M*N*K/np.median(ts)/1e9:6.0f}")
    print(name, "  ".join(out), flush=True)
