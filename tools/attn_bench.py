"""Attention kernel variants (waves per workgroup): time per launch and agreement with the 4-wave form."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load(); lib.llada_attn_force_variant.argtypes = [C.c_int]
H = 32; d = H * 128
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(8, 384), (1, 384), (8, 640), (2, 1152), (32, 384)]
for (B, L) in shapes:
    qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
    ref = None
    for v in (4, 2, 1):
        out = torch.zeros(B * L, d, device="cuda", dtype=torch.bfloat16)
        lib.llada_attn_force_variant(v)
        def call(): lib.llada_attn_bidir_bf16(qkv.data_ptr(), None, out.data_ptr(), B, L, H, None)
        call(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        err = (out.float() - ref.float()).abs().max().item()
        ts = []
        for r in range(3):
            for _ in range(3): call()
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        t = np.median(ts)
        print(f"B={B} L={L} variant {v:3d}: {t:7.1f} us  {4.0*B*H*L*L*128/t*1e-6:6.0f} TF  maxdiff {err:.4f}", flush=True)
lib.llada_attn_force_variant(0)
