"""Small-M GEMM sweep with COLD weights (a ring of weight matrices larger than the Infinity Cache, as in
situ where every layer has its own): tile rows x split count, time per launch in microseconds."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
Ms = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 256, 384]
import ctypes
lib.llada_gemm_assume_tiled.argtypes = [ctypes.c_int]
lib.llada_gemm_assume_tiled(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib.llada_gemm_use_pipelined(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
RING = 6
for M in Ms:
    for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (24576, 4096), "down": (4096, 12288)}.items():
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(RING if N * K * 2 > 60e6 else 4 * RING)]
        Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
        out = []
        for rows in (128, 192, 256):
            for S in (0, 1, 2, 3, 4, 6, 8, 12, 16):
                if S > 1 and (K // 64) % S: continue
                lib.llada_gemm_force_tile_rows(rows); lib.llada_gemm_force_splits(S if S else 0); ts = []
                wsp, wsb = (ws.data_ptr(), ws.numel() * 4) if S != 1 else (None, 0)
                def call(i):
                    lib.llada_gemm_bf16(A.data_ptr(), Ws[i % len(Ws)].data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, wsp, wsb, None)
                for r in range(3):
                    for i in range(3): call(i)
                    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); n = 24
                    e0.record()
                    for i in range(n): call(i)
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
                out.append(f"r{rows}/S{S if S else 'auto'}:{np.median(ts):6.1f}")
        print(f"M={M} {name} (W {N*K*2/1e6:.0f} MB, {N*K*2/8e6:.0f} us at 8 TB/s, {2*M*N*K/2.5e9:.0f} us at 2.5 PF)", " ".join(out), flush=True)
        del Ws
lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_force_splits(0)
