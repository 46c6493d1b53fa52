"""In-process A/B of the attention kernel of two builds of the library (interleaved rounds, same buffers):
python tools/ab_attn.py ab/libllada_hip_base.so llada_amd/libllada_hip.so [BxL ...]"""
import ctypes as C
import sys

import numpy as np
import torch

paths = sys.argv[1:3]
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[3:]] or [(8, 384), (1, 384), (8, 640), (2, 1152)]
libs = [C.CDLL(p) for p in paths]
H = 32
d = H * 128
for (B, L) in shapes:
    qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
    outs = [torch.zeros(B * L, d, device="cuda", dtype=torch.bfloat16) for _ in libs]
    st = torch.cuda.current_stream().cuda_stream

    def call(i):
        rc = libs[i].llada_attn_bidir_bf16(C.c_void_p(qkv.data_ptr()), None, C.c_void_p(outs[i].data_ptr()), B, L, H, C.c_void_p(st))
        assert rc == 0
    ts = [[] for _ in libs]
    for rnd in range(5):
        for i in range(len(libs)):
            for _ in range(3):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1) / 30 * 1e3)
    same = torch.equal(outs[0], outs[1]) if len(libs) == 2 else None
    flops = 4.0 * B * H * L * L * 128
    print(f"B={B} L={L}: " + "  ".join(f"{paths[i].split('/')[-1]} {np.median(ts[i]):7.1f} us (min {min(ts[i]):.1f}, {flops / np.median(ts[i]) * 1e-6:5.0f} TF)"
                                        for i in range(len(libs))) + f"  identical={same}", flush=True)
