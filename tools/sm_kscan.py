"""Fixed cost and cost per 64-deep K step of the small-M GEMM loop: time N = 12288 (rotary epilogue), 24576 (SwiGLU)
and 4096 (plain) at M = 384 for several K; the slope is the K loop, the intercept launch + fill + epilogue."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
M = int(sys.argv[3]) if len(sys.argv) > 3 else 384
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # extra elements per row of A (lda = K + PAD)
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # diagnostic bits of the small-M loop: 4 no DMA, 8 no LDS reads, 16 no MFMA
lib.llada_gemm_no_store.argtypes = [__import__("ctypes").c_int]
st = torch.cuda.current_stream().cuda_stream
cos = torch.randn(M, 64, device="cuda"); sin = torch.randn(M, 64, device="cuda")
cases = [("qkv", 12288, 2, (3, 4)), ("gateup", 24576, 3, (6,)), ("plain4096", 4096, 0, (1, 2)), ("plain12288", 12288, 0, (3,))]
if BITS and False:
    cases = [("plain12288", 12288, 0, (3,))]
for name, N, epi, ngs in cases:
    for ng in ngs + (0,):
        line = f"{name} N={N} " + (f"small-M 192x{32 * ng}" if ng else "main kernels (rows 192)") + ":"
        xs, ys = [], []
        for K in (1024, 2048, 4096, 8192):
            Ws = [tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(3)]
            A = torch.randn(M, K + PAD, device="cuda").to(torch.bfloat16)
            out_cols = N // 2 if epi == 3 else N
            out = torch.empty(M, out_cols, device="cuda", dtype=torch.bfloat16)
            def call(i):
                rc = lib.llada_gemm_bf16(A.data_ptr(), Ws[i % 3].data_ptr(), out.data_ptr(), M, N, K, K + PAD, out_cols, epi | 0x100, None, 0,
                                         cos.data_ptr() if epi == 2 else None, sin.data_ptr() if epi == 2 else None, M if epi == 2 else 0, None, 0, st)
                assert rc == 0, lib.llada_last_error()
            if ng:
                lib.llada_gemm_force_small_m(ng); lib.llada_gemm_no_store(0)
            else:
                lib.llada_gemm_force_tile_rows(192)
            ts = []
            for rnd in range(3):
                for i in range(3): call(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(24): call(i)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 24 * 1e3)
            lib.llada_gemm_force_small_m(0); lib.llada_gemm_force_tile_rows(0); lib.llada_gemm_no_store(0)
            xs.append(K / 64); ys.append(np.median(ts))
            line += f"  K={K}: {ys[-1]:.1f}"
            del Ws
        slope, icpt = np.polyfit(xs, ys, 1)
        print(line + f"   -> {slope:.3f} us per K step + {icpt:.1f} us", flush=True)
