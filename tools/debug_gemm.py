"""Debug helper: run the GEMM on small integer data and print which 16x16 output blocks are wrong."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mode = sys.argv[5] if len(sys.argv) > 5 else "rand"
g = np.random.Generator(np.random.PCG64(1))
if mode == "ktile":      # A = 1, W[n, k] = 1 only inside K tile j -> output counts how often tile j was accumulated
    j = int(sys.argv[6])
    A = np.ones((M, K), np.float32)
    W = np.zeros((N, K), np.float32); W[:, 64 * j:64 * (j + 1)] = 1
else:
    A = g.integers(-3, 4, size=(M, K)).astype(np.float32)
    W = g.integers(-3, 4, size=(N, K)).astype(np.float32)
At = torch.from_numpy(A).cuda().to(torch.bfloat16); Wt = torch.from_numpy(W).cuda().to(torch.bfloat16)
out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
lib.llada_gemm_force_tile_rows(rows)
rc = lib.llada_gemm_bf16(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, None)
torch.cuda.synchronize()
assert rc == 0, lib.llada_last_error()
o = out.float().cpu().numpy(); ref = A @ W.T
bad = (o != ref)
print("mismatch fraction", bad.mean())
if mode == "ktile":
    vals, cnt = np.unique(o, return_counts=True); print("values", dict(zip(vals.tolist(), cnt.tolist())))
bm = bad.reshape(M // 16, 16, N // 16, 16).any(axis=(1, 3))
for r in range(bm.shape[0]):
    print("".join("X" if v else "." for v in bm[r]))
