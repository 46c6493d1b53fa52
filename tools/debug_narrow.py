"""Debug: where does the 128-column GEMM tile differ from the 256-column one?"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from llada_amd import _lib
from oracle.bf16 import round_bf16
lib = _lib.load()
g = np.random.Generator(np.random.PCG64(1))
M, N, K = 350, 1024, 1024
A = torch.from_numpy(g.integers(-3, 4, size=(M, K)).astype(np.float32)).cuda().to(torch.bfloat16)
W = torch.from_numpy(g.integers(-3, 4, size=(N, K)).astype(np.float32)).cuda().to(torch.bfloat16)
ref = (A.float() @ W.float().T)
st = torch.cuda.current_stream().cuda_stream
for rows in (128, 192):
    for cols in (256, 128):
        lib.llada_gemm_force_tile_rows(rows); lib.llada_gemm_force_tile_cols(cols)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, K, N, 0, None, 0, None, None, 0, None, 0, st)
        torch.cuda.synchronize()
        bad = (out.float() != ref.to(torch.bfloat16).float())
        print(rows, cols, rc, "bad", int(bad.sum()), "nan", int(torch.isnan(out.float()).sum()))
        if bad.any():
            cols_bad = torch.nonzero(bad.any(dim=0)).reshape(-1).cpu().numpy()
            rows_bad = torch.nonzero(bad.any(dim=1)).reshape(-1).cpu().numpy()
            print("  bad cols", cols_bad[:40], "... n", len(cols_bad))
            print("  bad rows", rows_bad[:20], "... n", len(rows_bad))

print("---- random data, all epilogues ----")
from llada_amd.weights import pack_gate_up, pack_qkv
from oracle import model as omodel
def bf(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda().to(torch.bfloat16)
K = 1024
n_seq, L, H, F = 2, 175, 4, 768
M, d = n_seq * L, H * 128
A = bf(round_bf16(g.standard_normal((M, K), dtype=np.float32)))
Wp = bf(round_bf16(g.standard_normal((1024, K), dtype=np.float32) * 0.05))
R = bf(round_bf16(g.standard_normal((M, 1024), dtype=np.float32)))
Wq, Wk, Wv = (bf(round_bf16(g.standard_normal((d, K), dtype=np.float32) * 0.05)) for _ in range(3))
Wg, Wu = (bf(round_bf16(g.standard_normal((F, K), dtype=np.float32) * 0.05)) for _ in range(2))
cos, sin = omodel.rope_tables(L, 128, 500000.0)
ct, sn = torch.from_numpy(cos).cuda(), torch.from_numpy(sin).cuda()
qkv_w, gu_w = pack_qkv(Wq, Wk, Wv, H), pack_gate_up(Wg, Wu)
def gemm(W, epi, out_cols, R=None, rope=False):
    out = torch.full((M, out_cols), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, W.shape[0], K, K, out_cols, epi,
                             None if R is None else R.data_ptr(), 0 if R is None else R.shape[1],
                             ct.data_ptr() if rope else None, sn.data_ptr() if rope else None, L if rope else 0, None, 0, st)
    assert rc == 0, lib.llada_last_error()
    torch.cuda.synchronize()
    return out
for rows in (128, 192):
    lib.llada_gemm_force_tile_rows(rows)
    res = {}
    for cols in (256, 128):
        lib.llada_gemm_force_tile_cols(cols)
        res[cols] = (gemm(Wp, 0, 1024), gemm(Wp, 1, 1024, R=R), gemm(qkv_w, 2, 3 * d, rope=True), gemm(gu_w, 3, F))
    for name, a, b in zip(("plain", "residual", "rope", "swiglu"), res[256], res[128]):
        bad = a.float() != b.float()
        bad |= torch.isnan(a.float()) != torch.isnan(b.float())
        print(rows, name, "differ", int(bad.sum()), "nan wide", int(torch.isnan(a.float()).sum()), "nan narrow", int(torch.isnan(b.float()).sum()))
        if bad.any():
            cb = torch.nonzero(bad.any(dim=0)).reshape(-1).cpu().numpy()
            rb = torch.nonzero(bad.any(dim=1)).reshape(-1).cpu().numpy()
            print("   cols", cb[:48], len(cb), " rows", rb[:10], len(rb))
            i, j = int(rb[0]), int(cb[0])
            print("   e.g.", a[i, j].item(), b[i, j].item())
