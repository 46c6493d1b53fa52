"""What a small-M GEMM would gain if its weight rows were L2 hits: the QKV and gate/up shapes at M = 384 with K so small
that the whole weight matrix stays in the XCDs' L2s (one copy, re-used every launch: hot) against a ring of copies
(cold); the slope over K is the cost of a K step with the weights coming from the L2 / from HBM."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
M = 384
st = torch.cuda.current_stream().cuda_stream
cos = torch.randn(M, 64, device="cuda"); sin = torch.randn(M, 64, device="cuda")
shapes = (("qkv", 12288, 2, 192, 128), ("gateup", 24576, 3, 192, 256))
if len(sys.argv) > 1:   # a narrow matrix whose weights fit in EVERY XCD's L2 (the block -> XCD assignment may rotate between launches)
    shapes = (("narrow", int(sys.argv[1]), 0, 192, 128), ("narrow", int(sys.argv[1]), 0, 192, 256))
KS = (256, 512, 768, 1024) if len(sys.argv) <= 1 else (256, 384, 512)
for name, N, epi, rows, cols in shapes:
    for hot in (True, False):
        xs, ys = [], []
        line = f"{name} N={N} tile {rows}x{cols} weights {'hot (L2)' if hot else 'cold (HBM)'}:"
        for K in KS:
            NR = 1 if hot else max(2, int(600e6 / (N * K * 2)))
            Ws = [tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(NR)]
            A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            out_cols = N // 2 if epi == 3 else N
            out = torch.empty(M, out_cols, device="cuda", dtype=torch.bfloat16)
            def call(i):
                rc = lib.llada_gemm_bf16(A.data_ptr(), Ws[i % NR].data_ptr(), out.data_ptr(), M, N, K, K, out_cols, epi | 0x100, None, 0,
                                         cos.data_ptr() if epi == 2 else None, sin.data_ptr() if epi == 2 else None, M if epi == 2 else 0, None, 0, st)
                assert rc == 0, lib.llada_last_error()
            lib.llada_gemm_force_tile_rows(rows); lib.llada_gemm_force_tile_cols(cols)
            ts = []
            for rnd in range(3):
                for i in range(NR + 2): call(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(48): call(i)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 48 * 1e3)
            lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_force_tile_cols(0)
            xs.append(K / 64); ys.append(np.median(ts))
            line += f"  K={K}: {ys[-1]:.1f}"
            del Ws
        slope, icpt = np.polyfit(xs, ys, 1)
        print(line + f"   -> {slope:.3f} us per K step + {icpt:.1f} us", flush=True)
