"""Per-wave timeline of the attention kernel's key-tile loop from a diagnostic build (tools/build_variant.sh out.so
-DLLADA_ATTN_TRACE=1, loaded through LLADA_HIP_LIB): shader-clock stamps of workgroup 0, printed as average cycles per
segment: [loop top -> own DMA landed] [-> barrier passed] [-> QK MFMAs issued] [-> softmax done] [-> PV MFMAs issued]."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 384
H = 32; d = H * 128
st = torch.cuda.current_stream().cuda_stream
qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
out = torch.zeros(B * L, d, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(max(1 << 20, lib.llada_attn_workspace_bytes(B, L, L, H) // 4), dtype=torch.int32, device="cuda")
lib.llada_attn_force_key_splits(1)      # the workspace only carries the stamps
for _ in range(3):
    rc = lib.llada_attn_bidir_queries_bf16(qkv.data_ptr(), 3 * d, L, qkv.data_ptr(), None, out.data_ptr(), B, L, H, None, 0,
                                           ws.data_ptr(), ws.numel() * 4, st)
    assert rc == 0, lib.llada_last_error()
torch.cuda.synchronize()
nt = (L + 63) // 64
tr = ws[:4 * 32 * 8].cpu().numpy().astype(np.int64).reshape(4, 32, 8)[:, :nt, :6]
t0 = tr[:, 0, 0].min()
tr = (tr - t0) & 0xFFFFFFFF
names = ["dma wait", "barrier", "QK", "softmax", "PV"]
print("wave: " + "  ".join(f"{n:>9s}" for n in names) + "   | tile period")
for w in range(4):
    seg = [np.mean(tr[w, 1:, i + 1] - tr[w, 1:, i]) for i in range(5)]
    per = np.mean(np.diff(tr[w, :, 0])) if nt > 1 else 0
    print(f"  {w}: " + "  ".join(f"{x:9.0f}" for x in seg) + f"   | {per:.0f}")
print("wave 0, cycles since start, per tile:")
for t in range(nt):
    print("   ", t, tr[0, t].tolist())
