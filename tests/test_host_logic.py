"""Host-side logic of the drop-ins that needs no GPU: stratified mask counts, the host Philox
twin, weight packing, row sharding."""
import numpy as np
import torch

from oracle import likelihood as oll
from oracle import rng as orng
from tests.golden_util import load


def test_stratified_counts_match_torch_table():
    from llada_amd.get_log_likelihood import stratified_counts
    z, _ = load("linspace.npz")
    off = 0
    for T, b, k in zip(z["T"], z["b"], z["k"]):
        got = stratified_counts(int(k), int(b), int(T))
        assert np.array_equal(got, z["x"][off: off + b]), (T, b, k)
        assert np.array_equal(got, oll.stratified_counts(int(k), int(b), int(T)))
        off += b


def test_host_philox_matches_oracle_stream():
    from llada_amd import rng
    for seed in (0, 7, 2 ** 40 + 5):
        for it in range(5):
            for T in (1, 7, 128):
                assert rng.ll_k(seed, it, T) == orng.ll_k(seed, it, T)
    # full 4-word agreement on one counter
    k0, k1 = rng.key(123456789012345, rng.STREAM_GUMBEL)
    a = rng.philox4x32(5, 6, 7, 8, k0, k1)
    b = orng.philox4x32(np.uint64(5), np.uint64(6), np.uint64(7), np.uint64(8), k0, k1)
    assert list(a) == [int(v) for v in b]


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    from llada_amd import rng
    assert rng.philox4x32(0, 0, 0, 0, 0, 0) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert rng.philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == \
        (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert rng.philox4x32(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_weight_packing():
    from llada_amd.weights import head_permutation, pack_gate_up, pack_qkv
    perm = head_permutation()
    assert sorted(perm.tolist()) == list(range(128))
    # fragment pairs (2i, 2i+1) of 16 rows hold rotary partners j and j + 64
    for i in range(4):
        a, b = perm[32 * i: 32 * i + 16], perm[32 * i + 16: 32 * i + 32]
        assert np.array_equal(b, a + 64)
    d, H, F = 256, 2, 64
    q = torch.arange(d * d, dtype=torch.float32).reshape(d, d)
    packed = pack_qkv(q, q + 1e6, q + 2e6, H)
    assert packed.shape == (3 * d, d)
    assert torch.equal(packed[:128], q[torch.from_numpy(perm)])
    assert torch.equal(packed[128:256], q[128 + torch.from_numpy(perm)])
    assert torch.equal(packed[2 * d:], q + 2e6)
    g = torch.arange(F * 8, dtype=torch.float32).reshape(F, 8)
    gu = pack_gate_up(g, -g)
    assert torch.equal(gu[:16], g[:16]) and torch.equal(gu[16:32], -g[:16]) and torch.equal(gu[32:48], g[16:32])


def test_shard_bounds():
    from llada_amd.parallel import shard_bounds
    assert shard_bounds(64, 8) == [0, 8, 16, 24, 32, 40, 48, 56, 64]
    assert shard_bounds(3, 2) == [0, 2, 3]
    assert shard_bounds(1, 4) == [0, 1, 1, 1, 1]


def test_config_counts():
    from llada_amd import LLaDAConfig
    assert LLaDAConfig().linear_params() == 7497318400


def test_tile_major_layout_roundtrip_and_index_formula():
    """weights.tile_major: element (n, k) of an [N, K] matrix moves to
    ((n // 256 * (K // 64) + k // 64) * 256 + n % 256) * 64 + k % 64 (what LLADA_GEMM_W_TILED addresses)."""
    import torch

    from llada_amd.weights import LLaDAConfig, can_tile, from_tile_major, tile_major
    N, K = 512, 192
    w = torch.arange(N * K, dtype=torch.int32).reshape(N, K)
    t = tile_major(w)
    assert t.shape == w.shape and torch.equal(from_tile_major(t), w)
    flat = t.reshape(-1)
    for n, k in ((0, 0), (1, 0), (0, 64), (255, 63), (256, 0), (300, 130), (511, 191)):
        at = ((n // 256 * (K // 64) + k // 64) * 256 + n % 256) * 64 + k % 64
        assert int(flat[at]) == n * K + k
    assert can_tile(LLaDAConfig()) and can_tile(LLaDAConfig(d_model=256, n_heads=2, mlp_hidden=512, vocab_size=512))
    assert not can_tile(LLaDAConfig(d_model=256, n_heads=2, mlp_hidden=192, vocab_size=512))
