"""Per-kernel parity of the forward's building blocks (through the C ABI) against numpy
restatements / the oracle, on a real MI355X."""
import ctypes as C

import numpy as np
import pytest

from oracle import model as omodel
from oracle.bf16 import round_bf16, ulp_distance

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def bf(a):
    return dev(a.astype(np.float32), torch.bfloat16)


def host(t):
    return t.float().cpu().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(lib, rc):
    assert rc == 0, lib.llada_last_error().decode()


def gemm(lib, A, W, epi=0, R=None, cos=None, sin=None, seq_len=0, out_cols=None, workspace=None):
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, out_cols or N), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, K, out.shape[1], epi,
                             None if R is None else R.data_ptr(), 0 if R is None else R.shape[1],
                             None if cos is None else cos.data_ptr(), None if sin is None else sin.data_ptr(),
                             seq_len, None if workspace is None else workspace.data_ptr(),
                             0 if workspace is None else workspace.numel() * workspace.element_size(), stream())
    check(lib, rc)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("tile_rows", [0, 128, 192, 256])
@pytest.mark.parametrize("shape", [(256, 256, 64), (300, 512, 128), (1000, 768, 256), (37, 264, 192), (777, 1024, 512),
                                   (1111, 520, 1024), (400, 256, 384)])
def test_gemm_exact_integers(lib, shape, tile_rows):
    """Small-integer operands make every float32 partial sum exact: the result must be bit-exact
    whatever the tile shape, which pins the MFMA fragment maps, the LDS swizzle and the bounds."""
    M, N, K = shape
    g = np.random.Generator(np.random.PCG64(M * 7 + N))
    A = g.integers(-3, 4, size=(M, K)).astype(np.float32)
    W = g.integers(-3, 4, size=(N, K)).astype(np.float32)
    lib.llada_gemm_force_tile_rows(tile_rows)
    try:
        out = host(gemm(lib, bf(A), bf(W)))
    finally:
        lib.llada_gemm_force_tile_rows(0)
    ref = round_bf16(A @ W.T)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("tile_rows", [128, 192, 256])
def test_gemm_pipelined_equals_simple(lib, tile_rows):
    """The quarter-tile pipelined main loop, its persistent form (several tiles per workgroup as one
    continuous K stream; 8 and 24 workgroups over ~28-32 tiles) and the simple two-stage loop
    accumulate every output in the same k order: bit-identical results on random data, ragged M and
    N, long K, for every epilogue."""
    from llada_amd.weights import pack_gate_up, pack_qkv
    M, N, K = 1000, 776, 2048
    g = np.random.Generator(np.random.PCG64(21))
    A = bf(round_bf16(g.standard_normal((M, K), dtype=np.float32)))
    W = bf(round_bf16(g.standard_normal((N, K), dtype=np.float32) * 0.05))
    R = bf(round_bf16(g.standard_normal((M, N), dtype=np.float32)))
    # fused-epilogue operands: 2 heads of 128 (rope), 384 hidden units (SwiGLU), K = 256 * 8
    n_seq, L, H, F = 4, 250, 2, 384
    d = H * 128
    A2 = bf(round_bf16(g.standard_normal((n_seq * L, K), dtype=np.float32)))
    Wq, Wk, Wv = (bf(round_bf16(g.standard_normal((d, K), dtype=np.float32) * 0.05)) for _ in range(3))
    Wg, Wu = (bf(round_bf16(g.standard_normal((F, K), dtype=np.float32) * 0.05)) for _ in range(2))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    ct, st = dev(cos), dev(sin)
    qkv_w, gu_w = pack_qkv(Wq, Wk, Wv, H), pack_gate_up(Wg, Wu)

    def run_all():
        return (gemm(lib, A, W), gemm(lib, A, W, epi=1, R=R),
                gemm(lib, A2, qkv_w, epi=2, cos=ct, sin=st, seq_len=L), gemm(lib, A2, gu_w, epi=3, out_cols=F))

    lib.llada_gemm_force_tile_rows(tile_rows)
    try:
        lib.llada_gemm_use_pipelined(0)
        slow = run_all()
        lib.llada_gemm_use_pipelined(3)
        fast = run_all()
        lib.llada_gemm_use_pipelined(2)
        streams = []
        for grid in (8, 24, 256):
            lib.llada_gemm_stream_grid(grid)
            streams.append(run_all())
    finally:
        lib.llada_gemm_stream_grid(256)
        lib.llada_gemm_use_pipelined(1)
        lib.llada_gemm_force_tile_rows(0)
    for a, b in zip(slow, fast):
        assert torch.equal(a, b)
    for st_ in streams:
        for a, b in zip(slow, st_):
            assert torch.equal(a, b)


@pytest.mark.parametrize("splits", [1, 2])
@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("tile_rows", [128, 192])
def test_gemm_narrow_tile_equals_wide(lib, tile_rows, tiled, splits):
    """The 128-column tile of the pipelined loop (used when the 256-column one would leave most CUs idle: one
    sequence) accumulates every output in the same k order as the 256-column tile: bit-identical results on random
    data for every epilogue, ragged M, row-major and tile-major weights, with and without split-K (the split count
    is forced so that both tiles cut K the same way)."""
    from llada_amd.weights import pack_gate_up, pack_qkv, tile_major
    K = 1024
    g = np.random.Generator(np.random.PCG64(77))
    n_seq, L, H, F = 2, 175, 4, 768
    M, d = n_seq * L, H * 128
    A = bf(round_bf16(g.standard_normal((M, K), dtype=np.float32)))
    Wp = bf(round_bf16(g.standard_normal((1024, K), dtype=np.float32) * 0.05))
    R = bf(round_bf16(g.standard_normal((M, 1024), dtype=np.float32)))
    Wq, Wk, Wv = (bf(round_bf16(g.standard_normal((d, K), dtype=np.float32) * 0.05)) for _ in range(3))
    Wg, Wu = (bf(round_bf16(g.standard_normal((F, K), dtype=np.float32) * 0.05)) for _ in range(2))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    ct, st = dev(cos), dev(sin)
    qkv_w, gu_w = pack_qkv(Wq, Wk, Wv, H), pack_gate_up(Wg, Wu)
    flag = 0
    if tiled:
        Wp, qkv_w, gu_w, flag = tile_major(Wp), tile_major(qkv_w), tile_major(gu_w), 0x100
    ws = torch.empty(4 * M * 3 * d, dtype=torch.float32, device="cuda")

    def run_all():
        return (gemm(lib, A, Wp, epi=0 | flag, workspace=ws), gemm(lib, A, Wp, epi=1 | flag, R=R, workspace=ws),
                gemm(lib, A, qkv_w, epi=2 | flag, cos=ct, sin=st, seq_len=L, workspace=ws),
                gemm(lib, A, gu_w, epi=3 | flag, out_cols=F, workspace=ws))

    lib.llada_gemm_force_tile_rows(tile_rows)
    lib.llada_gemm_force_splits(splits)
    try:
        lib.llada_gemm_force_tile_cols(256)
        wide = run_all()
        lib.llada_gemm_force_tile_cols(128)
        narrow = run_all()
    finally:
        lib.llada_gemm_force_tile_cols(0)
        lib.llada_gemm_force_splits(0)
        lib.llada_gemm_force_tile_rows(0)
    for a, b in zip(wide, narrow):
        assert not torch.isnan(a.float()).any() and torch.equal(a, b)


@pytest.mark.parametrize("splits", [1, 2])
@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("ng", [1, 2, 3, 4, 6])
def test_gemm_small_m_loop_exact_integers(lib, ng, tiled, splits):
    """The loop for one or two sequences (192 x 32*ng tiles, the two K halves of a step on different waves) adds in
    another order than the main kernels, so it is compared on small integers, where every order is exact: all
    four epilogues bit-identical to the 192 x 256 tile, ragged M, column tiles that overhang N and (ng = 3) that
    straddle the rotary / pass-through boundary, row-major and tile-major weights, with and without split-K."""
    from llada_amd.weights import pack_gate_up, pack_qkv, tile_major
    K = 1024
    g = np.random.Generator(np.random.PCG64(78))
    n_seq, L, H, F = 2, 175, 4, 768
    M, d = n_seq * L, H * 128
    ints = lambda *shape: bf(g.integers(-3, 4, size=shape).astype(np.float32))
    A, Wp = ints(M, K), ints(1024, K)
    R = bf(round_bf16(g.standard_normal((M, 1024), dtype=np.float32) * 10))
    Wq, Wk, Wv = (ints(d, K) for _ in range(3))
    Wg, Wu = (bf(g.integers(-1, 2, size=(F, K)).astype(np.float32) * 0.03125) for _ in range(2))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    ct, st = dev(cos), dev(sin)
    qkv_w, gu_w = pack_qkv(Wq, Wk, Wv, H), pack_gate_up(Wg, Wu)
    flag = 0
    if tiled:
        Wp, qkv_w, gu_w, flag = tile_major(Wp), tile_major(qkv_w), tile_major(gu_w), 0x100
    ws = torch.empty(4 * M * 3 * d, dtype=torch.float32, device="cuda")

    def run_all():
        return (gemm(lib, A, Wp, epi=0 | flag, workspace=ws), gemm(lib, A, Wp, epi=1 | flag, R=R, workspace=ws),
                gemm(lib, A, qkv_w, epi=2 | flag, cos=ct, sin=st, seq_len=L, workspace=ws),
                gemm(lib, A, gu_w, epi=3 | flag, out_cols=F, workspace=ws))

    lib.llada_gemm_force_splits(splits)
    try:
        lib.llada_gemm_force_tile_rows(192)
        lib.llada_gemm_force_tile_cols(256)
        wide = run_all()
        lib.llada_gemm_force_tile_rows(0)
        lib.llada_gemm_force_tile_cols(0)
        lib.llada_gemm_force_small_m(ng)
        small = run_all()
    finally:
        lib.llada_gemm_force_small_m(0)
        lib.llada_gemm_force_tile_cols(0)
        lib.llada_gemm_force_splits(0)
        lib.llada_gemm_force_tile_rows(0)
    for i, (a, b) in enumerate(zip(wide, small)):
        assert not torch.isnan(a.float()).any() and a.float().abs().max() > 0
        assert torch.equal(a, b), f"epilogue {i}: {(a.float() - b.float()).abs().max().item()}"


@pytest.mark.parametrize("splits", [2, 4, 8])
@pytest.mark.parametrize("tile_rows", [128, 192, 256])
def test_gemm_splitk_exact_integers(lib, splits, tile_rows):
    """Split-K (float32 partial tiles + ordered reduce) for every epilogue: exact on integers."""
    M, N, K = 300, 768, 1024
    g = np.random.Generator(np.random.PCG64(31))
    A = g.integers(-3, 4, size=(M, K)).astype(np.float32)
    W = g.integers(-3, 4, size=(N, K)).astype(np.float32)
    R = round_bf16(g.standard_normal((M, N), dtype=np.float32) * 10)
    ws = torch.empty(splits * M * N, dtype=torch.float32, device="cuda")
    lib.llada_gemm_force_tile_rows(tile_rows)
    lib.llada_gemm_force_splits(splits)
    try:
        out = host(gemm(lib, bf(A), bf(W), workspace=ws))
        res = host(gemm(lib, bf(A), bf(W), epi=1, R=bf(R), workspace=ws))
    finally:
        lib.llada_gemm_force_splits(0)
        lib.llada_gemm_force_tile_rows(0)
    assert np.array_equal(out, round_bf16(A @ W.T))
    assert np.array_equal(res, round_bf16(R + round_bf16(A @ W.T)))


def test_gemm_splitk_fused_epilogues(lib):
    """Split-K + RoPE / SwiGLU second pass == the single-pass fused epilogues (integer operands make
    the accumulation order irrelevant, so the comparison is bit-exact)."""
    from llada_amd.weights import pack_gate_up, pack_qkv
    n_seq, L, H = 2, 50, 2
    d, F = H * 128, 384
    M = n_seq * L
    g = np.random.Generator(np.random.PCG64(32))
    A = bf(g.integers(-2, 3, size=(M, d)).astype(np.float32))
    Wq, Wk, Wv = (bf(g.integers(-2, 3, size=(d, d)).astype(np.float32)) for _ in range(3))
    Wg, Wu = (bf(g.integers(-2, 3, size=(F, d)).astype(np.float32) * 0.125) for _ in range(2))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    ct, st = dev(cos), dev(sin)
    qkv_w, gu_w = pack_qkv(Wq, Wk, Wv, H), pack_gate_up(Wg, Wu)
    ws = torch.empty(4 * M * 3 * d, dtype=torch.float32, device="cuda")
    try:
        one_q = gemm(lib, A, qkv_w, epi=2, cos=ct, sin=st, seq_len=L)
        one_g = gemm(lib, A, gu_w, epi=3, out_cols=F)
        lib.llada_gemm_force_splits(2)
        two_q = gemm(lib, A, qkv_w, epi=2, cos=ct, sin=st, seq_len=L, workspace=ws)
        two_g = gemm(lib, A, gu_w, epi=3, out_cols=F, workspace=ws)
    finally:
        lib.llada_gemm_force_splits(0)
    assert torch.equal(one_q, two_q)
    assert torch.equal(one_g, two_g)


def test_gemm_splitk_auto_small_m(lib):
    """M = 384 (one sequence): the library splits K on its own when given a workspace; results stay
    within accumulation-order noise of the unsplit run."""
    M, N, K = 384, 4096, 4096
    g = np.random.Generator(np.random.PCG64(33))
    A = bf(round_bf16(g.standard_normal((M, K), dtype=np.float32)))
    W = bf(round_bf16(g.standard_normal((N, K), dtype=np.float32) * 0.02))
    need = lib.llada_gemm_workspace_bytes(M, N, K)
    assert need > 0 and lib.llada_gemm_workspace_bytes(3072, N, K) == 0
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    a = host(gemm(lib, A, W))
    b = host(gemm(lib, A, W, workspace=ws))
    assert ulp_distance(a, b).max() <= 1 or np.abs(a - b).max() <= 2e-2


def test_gather_rows(lib):
    g = np.random.Generator(np.random.PCG64(41))
    src = bf(round_bf16(g.standard_normal((300, 256), dtype=np.float32)))
    rows = torch.from_numpy(g.integers(0, 300, size=77).astype(np.int32)).cuda()
    out = torch.empty(77, 256, dtype=torch.bfloat16, device="cuda")
    check(lib, lib.llada_gather_rows_bf16(src.data_ptr(), rows.data_ptr(), out.data_ptr(), 77, 256, 300, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, src[rows.long()])


@pytest.mark.parametrize("shape", [(512, 512, 256), (3072, 1024, 4096), (130, 768, 1024)])
def test_gemm_random(lib, shape):
    M, N, K = shape
    g = np.random.Generator(np.random.PCG64(11))
    A = round_bf16(g.standard_normal((M, K), dtype=np.float32))
    W = round_bf16(g.standard_normal((N, K), dtype=np.float32) * 0.05)
    out = host(gemm(lib, bf(A), bf(W)))
    ref = (A.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
    err = np.abs(out - ref)
    assert (err <= np.abs(ref) * 2.0 ** -8 + 1e-4 * np.sqrt(K)).all()


def test_gemm_residual(lib):
    M, N, K = 300, 512, 128
    g = np.random.Generator(np.random.PCG64(5))
    A = g.integers(-3, 4, size=(M, K)).astype(np.float32)
    W = g.integers(-3, 4, size=(N, K)).astype(np.float32)
    R = round_bf16(g.standard_normal((M, N), dtype=np.float32) * 10)
    Rt = bf(R)
    out = host(gemm(lib, bf(A), bf(W), epi=1, R=Rt))
    ref = round_bf16(R + round_bf16(A @ W.T))
    assert np.array_equal(out, ref)
    # in place (C aliases R), as the forward uses it
    At, Wt = bf(A), bf(W)
    rc = lib.llada_gemm_bf16(At.data_ptr(), Wt.data_ptr(), Rt.data_ptr(), M, N, K, K, N, 1, Rt.data_ptr(), N,
                             None, None, 0, None, 0, stream())
    check(lib, rc)
    torch.cuda.synchronize()
    assert np.array_equal(host(Rt), ref)


@pytest.mark.parametrize("M,N,K,splits", [(96, 512, 1024, 4), (384, 4096, 4096, 0), (384, 4096, 12288, 0),
                                          (200, 8192, 512, 2), (1024, 512, 256, 0), (3072, 4096, 4096, 0)])
def test_gemm_residual_rmsnorm_fused(lib, M, N, K, splits):
    """Projection + residual + the norm that follows: the fused entry point (norm folded into the
    split-K second pass when there is one) equals the residual GEMM followed by the RMSNorm kernel
    bit for bit, and matches the oracle's arithmetic."""
    g = np.random.Generator(np.random.PCG64(M + N + K))
    A = bf(g.standard_normal((M, K), dtype=np.float32))
    W = bf(g.standard_normal((N, K), dtype=np.float32) * 0.05)
    X0 = bf(g.standard_normal((M, N), dtype=np.float32) * 2)
    nw = bf(1.0 + 0.1 * g.standard_normal(N, dtype=np.float32))
    ws = torch.empty(int(lib.llada_gemm_workspace_bytes(M, N, K)) or 16, dtype=torch.uint8, device="cuda")
    eps = 1e-5
    lib.llada_gemm_force_splits(splits)
    try:
        # reference: two kernels
        X1 = X0.clone()
        check(lib, lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), X1.data_ptr(), M, N, K, K, N, 1, X1.data_ptr(), N,
                                       None, None, 0, ws.data_ptr(), ws.numel(), stream()))
        H1 = torch.empty_like(X1)
        check(lib, lib.llada_rmsnorm_bf16(X1.data_ptr(), nw.data_ptr(), H1.data_ptr(), M, N, eps, None, stream()))
        # fused
        X2 = X0.clone()
        H2 = torch.full_like(X2, float("nan"))
        check(lib, lib.llada_gemm_residual_rmsnorm_bf16(A.data_ptr(), W.data_ptr(), X2.data_ptr(), H2.data_ptr(),
                                                        nw.data_ptr(), eps, M, N, K, K, 0, ws.data_ptr(), ws.numel(),
                                                        stream()))
        # the same with the weight stored tile-major
        from llada_amd.weights import tile_major
        Wt = tile_major(W)
        X3 = X0.clone()
        H3 = torch.full_like(X3, float("nan"))
        check(lib, lib.llada_gemm_residual_rmsnorm_bf16(A.data_ptr(), Wt.data_ptr(), X3.data_ptr(), H3.data_ptr(),
                                                        nw.data_ptr(), eps, M, N, K, K, 1, ws.data_ptr(), ws.numel(),
                                                        stream()))
        torch.cuda.synchronize()
    finally:
        lib.llada_gemm_force_splits(0)
    assert torch.equal(X1, X2) and torch.equal(H1, H2)
    assert torch.equal(X1, X3) and torch.equal(H1, H3)
    x = host(X2)
    # two roundings (x * rstd, then * w): a 1-ulp difference of the first can move the second by one more
    ref_h = omodel.rmsnorm(x, host(nw), eps, True)
    ud = ulp_distance(host(H2), ref_h)
    assert ud.max() <= 2 and (ud <= 1).mean() >= 0.999


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("tile_rows", [128, 192, 256])
@pytest.mark.parametrize("M,N,K,splits", [(300, 512, 256, 0), (1000, 768, 512, 0), (64, 1024, 1024, 4), (37, 256, 64, 0),
                                          (640, 2048, 384, 2)])
def test_gemm_tile_major_weights(lib, M, N, K, splits, tile_rows, mode):
    """Tile-major weight storage is a pure change of addressing: every main loop (simple, pipelined,
    persistent), tile height and split count gives the row-major result bit for bit."""
    from llada_amd.weights import tile_major
    g = np.random.Generator(np.random.PCG64(M * 7 + N + K))
    A = bf(g.standard_normal((M, K), dtype=np.float32))
    W = bf(g.standard_normal((N, K), dtype=np.float32) * 0.05)
    Wt = tile_major(W)
    ws = torch.empty(16 * M * N * 4, dtype=torch.uint8, device="cuda")
    lib.llada_gemm_force_tile_rows(tile_rows)
    lib.llada_gemm_use_pipelined(mode)
    lib.llada_gemm_force_splits(splits)
    try:
        out = []
        for w, flag in ((W, 0), (Wt, 0x100)):
            Cc = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            check(lib, lib.llada_gemm_bf16(A.data_ptr(), w.data_ptr(), Cc.data_ptr(), M, N, K, K, N, 0 | flag, None, 0,
                                           None, None, 0, ws.data_ptr(), ws.numel(), stream()))
            out.append(Cc)
        torch.cuda.synchronize()
    finally:
        lib.llada_gemm_force_tile_rows(0)
        lib.llada_gemm_use_pipelined(1)
        lib.llada_gemm_force_splits(0)
    assert torch.equal(out[0], out[1])
    ref = host(A) @ host(W).T
    assert np.abs(host(out[1]) - ref).max() <= 2e-2 * np.abs(ref).max()


def test_gemm_tile_major_rejects_ragged_n(lib):
    A = torch.zeros((16, 64), dtype=torch.bfloat16, device="cuda")
    W = torch.zeros((264, 64), dtype=torch.bfloat16, device="cuda")
    Cc = torch.zeros((16, 264), dtype=torch.bfloat16, device="cuda")
    rc = lib.llada_gemm_bf16(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), 16, 264, 64, 64, 264, 0x100, None, 0, None, None,
                             0, None, 0, stream())
    assert rc == -2 and b"tile-major" in lib.llada_last_error()


def test_gemm_rope_epilogue(lib):
    """Packed QKV projection + rotary embedding == natural projection followed by the oracle's rope."""
    from llada_amd.weights import head_permutation, pack_qkv
    n_seq, L, H = 2, 37, 2
    d = H * 128
    M = n_seq * L
    g = np.random.Generator(np.random.PCG64(9))
    A = round_bf16(g.standard_normal((M, d), dtype=np.float32))
    Wq, Wk, Wv = (round_bf16(g.standard_normal((d, d), dtype=np.float32) * 0.06) for _ in range(3))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    packed = pack_qkv(bf(Wq), bf(Wk), bf(Wv), H)
    out = host(gemm(lib, bf(A), packed, epi=2, cos=dev(cos), sin=dev(sin), seq_len=L))
    # exact float32 accumulation is order dependent: take the GPU's own plain projection as the input
    nat = host(gemm(lib, bf(A), bf(np.concatenate([Wq, Wk, Wv]))))
    q = nat[:, :d].reshape(n_seq, L, H, 128).transpose(0, 2, 1, 3)
    k = nat[:, d:2 * d].reshape(n_seq, L, H, 128).transpose(0, 2, 1, 3)
    qr = omodel.apply_rope(q, cos, sin, True).transpose(0, 2, 1, 3).reshape(M, H, 128)
    kr = omodel.apply_rope(k, cos, sin, True).transpose(0, 2, 1, 3).reshape(M, H, 128)
    perm = head_permutation()
    assert np.array_equal(out[:, :d].reshape(M, H, 128), qr[:, :, perm])
    assert np.array_equal(out[:, d:2 * d].reshape(M, H, 128), kr[:, :, perm])
    assert np.array_equal(out[:, 2 * d:], nat[:, 2 * d:])


def test_gemm_swiglu_epilogue(lib):
    from llada_amd.weights import pack_gate_up
    M, d, F = 200, 256, 384
    g = np.random.Generator(np.random.PCG64(10))
    A = round_bf16(g.standard_normal((M, d), dtype=np.float32))
    Wg, Wu = (round_bf16(g.standard_normal((F, d), dtype=np.float32) * 0.08) for _ in range(2))
    out = host(gemm(lib, bf(A), pack_gate_up(bf(Wg), bf(Wu)), epi=3, out_cols=F))
    nat = host(gemm(lib, bf(A), bf(np.concatenate([Wg, Wu]))))
    gate, up = nat[:, :F], nat[:, F:]
    ref = round_bf16(round_bf16(gate / (np.float32(1) + np.exp(-gate))) * up)
    assert ulp_distance(out, ref).max() <= 1
    # and the unfused kernel on the natural layout
    o2 = torch.empty((M, F), dtype=torch.bfloat16, device="cuda")
    natt = bf(nat)
    check(lib, lib.llada_swiglu_bf16(natt.data_ptr(), o2.data_ptr(), M, F, stream()))
    torch.cuda.synchronize()
    assert ulp_distance(host(o2), ref).max() <= 1


def test_gemm_rejects_bad_shapes(lib):
    a = torch.zeros(8, 96, dtype=torch.bfloat16, device="cuda")
    rc = lib.llada_gemm_bf16(a.data_ptr(), a.data_ptr(), a.data_ptr(), 8, 8, 96, 96, 8, 0, None, 0, None, None, 0, None, 0,
                             None)
    assert rc == -2 and b"K=" in lib.llada_last_error()


@pytest.mark.parametrize("n_seq,L,H,masked", [(1, 64, 1, False), (2, 100, 2, False), (2, 384, 2, False),
                                             (3, 200, 2, True), (1, 513, 1, True), (2, 31, 1, False)])
@pytest.mark.parametrize("waves", [0, 1, 2, 4])
def test_attention(lib, n_seq, L, H, masked, waves):
    """waves = waves per workgroup (0: the library's choice); every variant serves any length."""
    d = H * 128
    g = np.random.Generator(np.random.PCG64(L + H))
    qkv = round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))
    # spike a few keys so the running max jumps late in the sweep (exercises the rescale path)
    qkv[L // 2, d:2 * d] *= 4
    qkv[L - 1, d:2 * d] *= 6
    km = None
    if masked:
        km = np.ones((n_seq, L), dtype=np.uint8)
        for b in range(n_seq):
            km[b, : 3 + 17 * b] = 0
        km[0, L // 3: L // 3 + 70] = 0          # a whole 64-key tile masked
    out = torch.full((n_seq * L, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    kmt = None if km is None else dev(km)
    qt = bf(qkv)
    lib.llada_attn_force_variant(waves)
    try:
        check(lib, lib.llada_attn_bidir_bf16(qt.data_ptr(), None if kmt is None else kmt.data_ptr(), out.data_ptr(),
                                             n_seq, L, H, stream()))
        torch.cuda.synchronize()
    finally:
        lib.llada_attn_force_variant(0)
    x = qkv.reshape(n_seq, L, 3, H, 128)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    ref = omodel.attention(q, k, v, km, False).transpose(0, 2, 1, 3).reshape(n_seq * L, d)
    err = np.abs(host(out) - ref)
    assert np.isfinite(host(out)).all()
    assert err.max() <= 1.5e-2 * np.abs(ref).max(), err.max()


def test_attention_left_padding_and_rising_scores(lib):
    """A left-padded sequence whose first key tiles are masked completely (the running maximum stays -inf until the
    first real key), and keys whose scores grow steadily along the sequence, so that a row's maximum rises in many
    small steps (below the lazy-rescaling threshold of 2^8) and in a few large ones."""
    n_seq, L, H = 2, 320, 2
    d = H * 128
    g = np.random.Generator(np.random.PCG64(11))
    qkv = round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))
    ramp = np.linspace(0.25, 3.0, L, dtype=np.float32)
    for b in range(n_seq):
        qkv[b * L:(b + 1) * L, d:2 * d] *= ramp[:, None]
    qkv[L + 300, d:2 * d] *= 8                 # one late, very large key in the padded sequence
    qkv = round_bf16(qkv)
    km = np.ones((n_seq, L), dtype=np.uint8)
    km[1, :150] = 0                            # two whole 64-key tiles and part of a third
    out = torch.full((n_seq * L, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    kmt, qt = dev(km), bf(qkv)
    check(lib, lib.llada_attn_bidir_bf16(qt.data_ptr(), kmt.data_ptr(), out.data_ptr(), n_seq, L, H, stream()))
    torch.cuda.synchronize()
    x = qkv.reshape(n_seq, L, 3, H, 128)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    ref = omodel.attention(q, k, v, km, False).transpose(0, 2, 1, 3).reshape(n_seq * L, d)
    assert np.isfinite(host(out)).all()
    err = np.abs(host(out) - ref)
    assert err.max() <= 1.5e-2 * np.abs(ref).max(), err.max()


def test_attention_exact_uniform(lib):
    """All-equal scores: softmax is exactly uniform, so the output is the mean of V; with V
    holding small integers over a power-of-two number of keys the result is exact in bf16."""
    n_seq, L, H = 1, 128, 1
    d = 128
    g = np.random.Generator(np.random.PCG64(3))
    qkv = np.zeros((L, 3 * d), dtype=np.float32)
    qkv[:, 2 * d:] = g.integers(-8, 9, size=(L, d)) * 2.0
    out = torch.empty((L, d), dtype=torch.bfloat16, device="cuda")
    qt = bf(qkv)
    check(lib, lib.llada_attn_bidir_bf16(qt.data_ptr(), None, out.data_ptr(), n_seq, L, H, stream()))
    torch.cuda.synchronize()
    ref = round_bf16(np.repeat(qkv[:, 2 * d:].mean(axis=0, keepdims=True), L, axis=0))
    assert np.array_equal(host(out), ref)


def test_attention_v_layout(lib):
    """One-hot attention (one huge key per query) returns that key's V row exactly: pins the
    transposed V fragment map."""
    L, d = 192, 128
    g = np.random.Generator(np.random.PCG64(4))
    qkv = np.zeros((L, 3 * d), dtype=np.float32)
    target = g.permutation(L)
    # query i = e_(i % 128) scaled, key j = e_(j % 128) scaled: q_i . k_j large iff same residue;
    # break the residue tie with a second coordinate
    for i in range(L):
        qkv[i, i % 64] = 64.0
        qkv[i, 64 + i // 64] = 64.0
    for j in range(L):
        i = int(np.where(target == j)[0][0])
        qkv[j, d + i % 64] = 64.0
        qkv[j, d + 64 + i // 64] = 64.0
    qkv[:, 2 * d:] = round_bf16(g.standard_normal((L, d), dtype=np.float32))
    out = torch.empty((L, d), dtype=torch.bfloat16, device="cuda")
    qt = bf(qkv)
    check(lib, lib.llada_attn_bidir_bf16(qt.data_ptr(), None, out.data_ptr(), 1, L, 1, stream()))
    torch.cuda.synchronize()
    assert np.array_equal(host(out), qkv[target, 2 * d:])


@pytest.mark.parametrize("M,d", [(5, 256), (300, 4096), (17, 1024)])
def test_rmsnorm(lib, M, d):
    g = np.random.Generator(np.random.PCG64(d))
    x = round_bf16(g.standard_normal((M, d), dtype=np.float32) * 3)
    w = round_bf16(1 + 0.1 * g.standard_normal(d, dtype=np.float32))
    out = torch.empty((M, d), dtype=torch.bfloat16, device="cuda")
    xt, wt = bf(x), bf(w)
    check(lib, lib.llada_rmsnorm_bf16(xt.data_ptr(), wt.data_ptr(), out.data_ptr(), M, d, 1e-5, None, stream()))
    torch.cuda.synchronize()
    ref = omodel.rmsnorm(x, w, 1e-5, True)
    assert ulp_distance(host(out), ref).max() <= 1
    # gathered rows
    idx = np.array([M - 1, 0, M // 2], dtype=np.int32)
    out2 = torch.empty((3, d), dtype=torch.bfloat16, device="cuda")
    it = dev(idx)
    check(lib, lib.llada_rmsnorm_bf16(xt.data_ptr(), wt.data_ptr(), out2.data_ptr(), 3, d, 1e-5, it.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert np.array_equal(host(out2), host(out)[idx])


def test_rope_kernel(lib):
    n_seq, L, H = 2, 45, 2
    d = H * 128
    g = np.random.Generator(np.random.PCG64(8))
    qkv = round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))
    cos, sin = omodel.rope_tables(L, 128, 500000.0)
    t = bf(qkv)
    ct, st = dev(cos), dev(sin)
    check(lib, lib.llada_rope_bf16(t.data_ptr(), ct.data_ptr(), st.data_ptr(), n_seq, L, H, stream()))
    torch.cuda.synchronize()
    out = host(t)
    for part in range(2):
        a = qkv[:, part * d:(part + 1) * d].reshape(n_seq, L, H, 128).transpose(0, 2, 1, 3)
        ref = omodel.apply_rope(a, cos, sin, True).transpose(0, 2, 1, 3).reshape(n_seq * L, d)
        assert np.array_equal(out[:, part * d:(part + 1) * d], ref)
    assert np.array_equal(out[:, 2 * d:], qkv[:, 2 * d:])


def test_embed(lib):
    V, d, n = 1000, 256, 77
    g = np.random.Generator(np.random.PCG64(2))
    wte = round_bf16(g.standard_normal((V, d), dtype=np.float32))
    ids = g.integers(0, V, size=n, dtype=np.int64)
    out = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    wt, it = bf(wte), dev(ids)
    check(lib, lib.llada_embed_bf16(wt.data_ptr(), it.data_ptr(), out.data_ptr(), n, d, V, stream()))
    torch.cuda.synchronize()
    assert np.array_equal(host(out), wte[ids])


def test_gemm_rope_epilogue_with_explicit_positions(lib):
    """llada_forward_positions rotates token m with row positions[m] of the rotary tables.  On a one-layer model:
    explicit positions equal to the column index reproduce the default numbering bit for bit; a sequence that sits
    behind 5 masked padding columns and is numbered from its own first token (positions_from_mask) gives the logits
    of the bare sequence to bf16 noise, as does numbering by column (rotary attention depends on position
    differences only); positions outside the tables are clamped, never read out of bounds."""
    import torch
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    from llada_amd.model import positions_from_mask
    cfg = LLaDAConfig(d_model=256, n_layers=1, n_heads=2, mlp_hidden=512, vocab_size=512)
    model = LLaDAModel(LLaDAWeights.random(cfg, "cuda", seed=5, std=0.05, head_scale=4.0))
    g = torch.Generator().manual_seed(1)
    n, pad = 40, 5
    row = torch.randint(0, 500, (1, n), generator=g)
    ids = torch.cat([torch.full((1, pad), 9, dtype=torch.long), row], dim=1).cuda()
    mask = torch.cat([torch.zeros(1, pad, dtype=torch.long), torch.ones(1, n, dtype=torch.long)], dim=1).cuda()
    alone = model(row.cuda()).logits.float()[0]
    with_pos = model(ids, mask, position_ids=positions_from_mask(mask)).logits.float()[0, pad:]
    by_col = model(ids, mask).logits.float()[0, pad:]
    # explicit positions equal to the column index are the default numbering, bit for bit
    cols = torch.arange(pad + n, dtype=torch.int32)[None, :].cuda()
    assert torch.equal(model(ids, mask, position_ids=cols).logits, model(ids, mask).logits)
    scale = alone.abs().max()
    assert (with_pos - alone).abs().max() <= 2e-2 * scale
    assert (by_col - alone).abs().max() <= 2e-2 * scale
    # out-of-table positions are clamped, not read out of bounds
    wild = torch.full((1, pad + n), 10 ** 6, dtype=torch.int32).cuda()
    assert torch.isfinite(model(ids, mask, position_ids=wild).logits.float()).all()


@pytest.mark.parametrize("M", [32, 192, 384, 1536, 3072])
def test_gemm_describe_plan_matches_launch(lib, M):
    """The host-only plan hook (pinned for the LLaDA-8B shapes by tests/test_gemm_plans_cpu.py) predicts the tile, the
    split count and the persistent form that llada_gemm_bf16 then launches, for plain / rotary / SwiGLU / residual
    epilogues at the model's projection shapes, with and without a split-K workspace."""
    import ctypes as C
    d, F = 4096, 12288
    shapes = [(3 * d, d, 2), (d, d, 1), (2 * F, d, 3), (d, F, 1)]
    cos = torch.zeros(M, 64, device="cuda")
    for (N, K, epi) in shapes:
        A = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
        W = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
        R = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") if epi == 1 else None
        for ws_bytes in (0, 16 * M * 3 * d * 4):
            if ws_bytes > (1 << 30):
                ws_bytes = 0
            ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device="cuda") if ws_bytes else None
            want, got = (C.c_int32 * 4)(), (C.c_int32 * 4)()
            assert lib.llada_gemm_describe_plan(M, N, K, epi | 0x100, ws_bytes, want) == 0
            gemm(lib, A, W, epi=epi | 0x100, R=R, cos=cos if epi == 2 else None, sin=cos if epi == 2 else None,
                 seq_len=M if epi == 2 else 0, out_cols=N // 2 if epi == 3 else None, workspace=ws)
            lib.llada_gemm_last_plan(got)
            assert tuple(got) == tuple(want), (M, N, K, epi, ws_bytes, tuple(got), tuple(want))
