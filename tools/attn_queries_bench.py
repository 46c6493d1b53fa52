"""Query-subset attention (block-cached step): time per call by forced key-split count."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
H = 32; d = H * 128
for (B, nq, L) in [(1, 32, 384), (8, 32, 384), (2, 128, 1152), (8, 64, 640)]:
    kv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
    q = torch.randn(B * nq, 3 * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * nq, d, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(lib.llada_attn_workspace_bytes(B, nq, L, H), dtype=torch.uint8, device="cuda")
    res = []
    for S in (1, 0, 2, 3, 6, 9, 16):
        lib.llada_attn_force_key_splits(S)
        def call(): lib.llada_attn_bidir_queries_bf16(q.data_ptr(), 3 * d, nq, kv.data_ptr(), None, out.data_ptr(), B, L, H, None, 0, ws.data_ptr(), ws.numel(), None)
        ts = []
        for r in range(3):
            for _ in range(3): call()
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res.append(f"S{S if S else 'auto'}:{np.median(ts):6.1f}")
    print(f"n_seq={B} n_q={nq} L={L}:", " ".join(res), "us", flush=True)
lib.llada_attn_force_key_splits(0)
