"""Time per launch of the four layer GEMMs at M = 3072 (quarter-tile-phase loop, 192-row tile, ring of weights) for a
build given through LLADA_HIP_LIB: run once per -DLLADA_GEMM_DEBUG variant."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M, d, F = 3072, 4096, 12288
cos = torch.randn(M, 64, device="cuda"); sin = torch.randn(M, 64, device="cuda")
line = ""
for name, (N, K, epi) in {"qkv": (3 * d, d, 2), "o": (d, d, 1), "gateup": (2 * F, d, 3), "down": (d, F, 1)}.items():
    Ws = [tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(3)]
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    R = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out_cols = N // 2 if epi == 3 else N
    out = torch.empty(M, out_cols, device="cuda", dtype=torch.bfloat16)
    lib.llada_gemm_force_tile_rows(192); lib.llada_gemm_use_pipelined(3)
    def call(i):
        rc = lib.llada_gemm_bf16(A.data_ptr(), Ws[i % 3].data_ptr(), out.data_ptr(), M, N, K, K, out_cols, epi | 0x100, R.data_ptr() if epi == 1 else None,
                                 N if epi == 1 else 0, cos.data_ptr() if epi == 2 else None, sin.data_ptr() if epi == 2 else None,
                                 384 if epi == 2 else 0, None, 0, st)
        assert rc == 0
    ts = []
    for rnd in range(3):
        for i in range(3): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): call(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12 * 1e3)
    line += f"  {name} {np.median(ts):6.1f}"
    del Ws
print(line, flush=True)
