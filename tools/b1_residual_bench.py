"""The two residual projections of a layer at M = 384 (attention output K = 4096, MLP output K = 12288) through
llada_gemm_residual_rmsnorm_bf16, as the batch-1 forward calls them (split-K partial tiles + fused reduce / residual /
norm): time per call with cold weights (a ring of weight sets), and with the write-out of the partial tiles skipped
(diagnostic hook, results garbage) to see what the float32 slab costs."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
M, d, F = 384, 4096, 12288
torch.manual_seed(0)
for name, K in (("attn_out", d), ("mlp_out", F)):
    ring = [tile_major((torch.randn(d, K, device="cuda") * 0.02).to(torch.bfloat16)) for _ in range(24)]   # > Infinity Cache
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    X = torch.randn(M, d, device="cuda").to(torch.bfloat16)
    H = torch.empty_like(X)
    nw = torch.ones(d, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(16 * M * d, dtype=torch.float32, device="cuda")
    res = []
    for no_store in (0, 1):
        lib.llada_gemm_no_store(no_store)
        def call(i):
            rc = lib.llada_gemm_residual_rmsnorm_bf16(A.data_ptr(), ring[i % len(ring)].data_ptr(), X.data_ptr(), H.data_ptr(), nw.data_ptr(),
                                                      1e-5, M, d, K, K, 1, ws.data_ptr(), ws.numel() * 4, None)
            assert rc == 0, lib.llada_last_error()
        ts = []
        for r in range(5):
            for i in range(4): call(i)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24): call(i)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 24 * 1e3)
        res.append(f"{'no partial-tile stores' if no_store else 'as run'}: {np.median(ts):6.1f} us")
    lib.llada_gemm_no_store(0)
    print(f"M={M} {name} (K={K}): " + ", ".join(res), flush=True)
