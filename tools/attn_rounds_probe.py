"""Full attention by shape: time per call and per (workgroup item x key tile) against how full the rounds of 512
resident workgroups are (items = query blocks x heads x sequences).  python tools/attn_rounds_probe.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
H = 32; d = H * 128
shapes = [(2, 1024), (2, 1152), (2, 1280), (4, 1152), (8, 256), (8, 384), (8, 512), (16, 384), (8, 640), (8, 768), (64, 384)]
for (B, L) in shapes:
    qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
    def call(): lib.llada_attn_bidir_bf16(qkv.data_ptr(), None, out.data_ptr(), B, L, H, None)
    ts = []
    for r in range(5):
        for _ in range(3): call()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = float(np.median(ts))
    items = ((L + 127) // 128) * H * B
    tiles = (L + 63) // 64
    fl = 4.0 * L * L * 128 * H * B
    print(f"n_seq={B:3d} L={L:5d}: {t:7.1f} us  {fl / t * 1e-6:6.0f} TFLOP/s  items {items:5d} = {items / 512:5.2f} rounds x {tiles:2d} tiles"
          f"  -> {t / (items / 512 * tiles):5.2f} us per full-round tile", flush=True)
