import ctypes as C, sys, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); pass
M, N, K = 3072, int(sys.argv[1]), int(sys.argv[2]); pipe = int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
lib.llada_gemm_use_pipelined(pipe); lib.llada_gemm_force_tile_rows(192)
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(n): lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
torch.cuda.synchronize()
