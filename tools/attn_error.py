"""Error of the attention kernel against a float64 reference on random data (optionally for two builds):
python tools/attn_error.py [lib.so ...]"""
import ctypes as C
import sys

import torch

paths = sys.argv[1:] or ["llada_amd/libllada_hip.so"]
H, L, B = 4, 384, 2
d = H * 128
g = torch.Generator(device="cuda").manual_seed(5)
for scale in (1.0, 4.0):
    qkv = (torch.randn(B * L, 3 * d, device="cuda", generator=g) * scale).to(torch.bfloat16)
    q, k, v = (t.reshape(B, L, H, 128).transpose(1, 2).double() for t in qkv.split(d, dim=1))
    ref = torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, dim=-1) @ v
    ref = ref.transpose(1, 2).reshape(B * L, d)
    for p in paths:
        lib = C.CDLL(p)
        out = torch.zeros(B * L, d, device="cuda", dtype=torch.bfloat16)
        rc = lib.llada_attn_bidir_bf16(C.c_void_p(qkv.data_ptr()), None, C.c_void_p(out.data_ptr()), B, L, H,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        err = (out.double() - ref).abs()
        rms = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"input scale {scale}: {p.split('/')[-1]:28s} max |err| {err.max().item():.3e}  rms rel {rms:.3e}")
