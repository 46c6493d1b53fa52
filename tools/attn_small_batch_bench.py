"""Full self-attention of ONE or two sequences (the batch-1 denoising step): 4-wave workgroups leave most CUs idle
(96 workgroups at 1 x 384), so time per call by waves per workgroup (V) and forced key-split count (S)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
H = 32; d = H * 128
for (B, L) in [(1, 384), (1, 192), (2, 384), (1, 640), (1, 1152), (2, 1152)]:
    qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(lib.llada_attn_workspace_bytes(B, L, L, H), dtype=torch.uint8, device="cuda")
    res = []
    for V in (4, 2):
        for S in (1, 2, 3, 6):
            lib.llada_attn_force_variant(V); lib.llada_attn_force_key_splits(S)
            def call(): lib.llada_attn_bidir_queries_bf16(qkv.data_ptr(), 3 * d, L, qkv.data_ptr(), None, out.data_ptr(), B, L, H, None, 0, ws.data_ptr(), ws.numel(), None)
            ts = []
            for r in range(3):
                for _ in range(3): call()
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): call()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            res.append(f"V{V}S{S}:{np.median(ts):6.1f}")
    print(f"n_seq={B} L={L}:", " ".join(res), "us", flush=True)
lib.llada_attn_force_key_splits(0); lib.llada_attn_force_variant(0)
