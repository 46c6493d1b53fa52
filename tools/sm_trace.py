"""Per-wave timeline of the small-M GEMM loop from a diagnostic build (make CXXFLAGS+=-DLLADA_SM_TRACE=1, loaded
through LLADA_HIP_LIB): shader-clock stamps of workgroup 0 at four points of every K step -- own DMA share landed /
barrier passed / refill issued (kh = 0) / MFMAs issued -- printed as average cycles per segment and wave."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from llada_amd.weights import tile_major
lib = _lib.load()
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, N, K = 384, int(sys.argv[2]) if len(sys.argv) > 2 else 12288, 4096
st = torch.cuda.current_stream().cuda_stream
W = tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
W2 = tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
W3 = tile_major((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")
lib.llada_gemm_force_small_m(ng)
for Wx in (W, W2, W3, W):
    rc = lib.llada_gemm_bf16(A.data_ptr(), Wx.data_ptr(), out.data_ptr(), M, N, K, K, N, 0x100, None, 0, None, None, 0, ws.data_ptr(), ws.numel() * 4, st)
    assert rc == 0, lib.llada_last_error()
torch.cuda.synchronize()
tr = ws[:2048].cpu().numpy().astype(np.int64).reshape(8, 64, 4)
t0 = tr[:, 0, 0].min()
tr = (tr - t0) & 0xFFFFFFFF
print("wave: landed->barrier  barrier->issued  issued->mfma-issued  mfma-issued->next landed   | K step period")
for w in range(8):
    seg = [np.mean(tr[w, 4:60, 1] - tr[w, 4:60, 0]), np.mean(tr[w, 4:60, 2] - tr[w, 4:60, 1]), np.mean(tr[w, 4:60, 3] - tr[w, 4:60, 2]),
           np.mean(tr[w, 5:61, 0] - tr[w, 4:60, 3])]
    print(f"  {w}: " + "  ".join(f"{x:8.0f}" for x in seg) + f"   | {np.mean(np.diff(tr[w, 4:60, 0])):.0f}")
print("first steps of wave 0 (cycles since start):")
for t in range(0, 12):
    print("   ", t, tr[0, t].tolist(), " wave 4:", tr[4, t].tolist())
