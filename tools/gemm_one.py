"""Run one projection GEMM a few times (for rocprofv3 passes): gemm_one.py N K [epilogue] [launches] [M] [split-K workspace: 0/1] [tile-major W: 0/1] [small-M loop columns / 32]."""
import sys, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
N, K = int(sys.argv[1]), int(sys.argv[2])
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5
M = int(sys.argv[5]) if len(sys.argv) > 5 else 3072
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
tiled = len(sys.argv) > 7 and int(sys.argv[7])
if tiled:
    from llada_amd.weights import tile_major
    W = tile_major(W)
ldc = N // 2 if epi == 3 else N
Cc = torch.empty(M, ldc, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda") if len(sys.argv) > 6 and int(sys.argv[6]) else None
wsp, wsb = (ws.data_ptr(), ws.numel() * 4) if ws is not None else (None, 0)
if len(sys.argv) > 8: lib.llada_gemm_force_small_m(int(sys.argv[8]))
for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, ldc, epi | (0x100 if tiled else 0), None, 0, None, None, 0, wsp, wsb, None)
torch.cuda.synchronize()
