"""The four projections of one LLaDA-8B layer at M = 384 (one sequence): time per launch with the automatic plan,
with the 256-column tile forced, with the 128-column tile forced and with the small-M loop at every tile width
(cold weights: a ring of layers; second argument 1 = one copy, hot)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 384
d, F = 4096, 12288
st = torch.cuda.current_stream().cuda_stream
shapes = {"qkv": (3 * d, d, 2), "o": (d, d, 1), "gateup": (2 * F, d, 3), "down": (d, F, 1)}
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 6   # ring of weight copies: 6 exceeds the Infinity Cache (cold weights), 1 keeps them hot
cos = torch.randn(M, 64, device="cuda"); sin = torch.randn(M, 64, device="cuda")
for name, (N, K, epi) in shapes.items():
    Ws = [tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(NR)]
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out_cols = N // 2 if epi == 3 else N
    out = torch.empty(M, out_cols, device="cuda", dtype=torch.bfloat16)
    R = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    ws = torch.empty(lib.llada_gemm_workspace_bytes(M, N, K) + 16, dtype=torch.uint8, device="cuda")
    def call(i):
        rc = lib.llada_gemm_bf16(A.data_ptr(), Ws[i % NR].data_ptr(), out.data_ptr(), M, N, K, K, out_cols, epi | 0x100,
                                 R.data_ptr() if epi == 1 else None, N if epi == 1 else 0, cos.data_ptr() if epi == 2 else None,
                                 sin.data_ptr() if epi == 2 else None, M if epi == 2 else 0, ws.data_ptr(), ws.numel() - 16, st)
        assert rc == 0, lib.llada_last_error()
    def timed():
        ts = []
        for rnd in range(3):
            for i in range(NR): call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24): call(i)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 24 * 1e3)
        return np.median(ts)
    print(f"M={M} {name}: N={N} K={K}   automatic plan: {timed():6.1f} us", flush=True)
    for rows in (128, 192):
        for cols in (256, 128):
            line = f"   rows {rows} cols {cols}:"
            for splits in (0, 1, 2, 4, 8):
                lib.llada_gemm_force_tile_rows(rows); lib.llada_gemm_force_tile_cols(cols); lib.llada_gemm_force_splits(splits)
                line += f"  S={splits}: {timed():6.1f}"
            print(line, flush=True)
    lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_force_tile_cols(0); lib.llada_gemm_force_splits(0)
    for ng in (1, 2, 3, 4, 6):
        line = f"   small-M loop, 192 x {32 * ng:3d}:"
        for splits in (1, 2, 4, 8):
            lib.llada_gemm_force_small_m(ng); lib.llada_gemm_force_splits(splits)
            line += f"  S={splits}: {timed():6.1f}"
        print(line, flush=True)
    lib.llada_gemm_force_small_m(0); lib.llada_gemm_force_splits(0)
    del Ws
