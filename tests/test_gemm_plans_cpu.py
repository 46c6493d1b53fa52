"""Launch plans of the bf16 GEMM for the LLaDA-8B projections (the model the reference calls at generate.py:88,92),
read back through the host-only hook llada_gemm_describe_plan: the choices the measured sweeps under profiles/
(r2_small_m_plan_sweep*, r3_small_m_plan_sweep_m192 / _m256, r3_mid_m_plan_sweep) settled are pinned here, so a change
to the planner's cost model that moves one of them shows up without a GPU."""
import ctypes as C

import pytest

D, F, V = 4096, 12288, 126464
QKV, O, GATEUP, DOWN, LM_HEAD = (3 * D, D, 2), (D, D, 1), (2 * F, D, 3), (D, F, 1), (V, D, 0)


def plan(lib, M, shape, workspace=None):
    N, K, epi = shape
    if workspace is None:   # what llada_forward gives every projection: the largest request of the layer's shapes
        workspace = max(lib.llada_gemm_workspace_bytes(M, n, k) for (n, k, _) in (QKV, O, GATEUP, DOWN))
    out = (C.c_int32 * 4)()
    assert lib.llada_gemm_describe_plan(M, N, K, epi, workspace, out) == 0
    return tuple(out)


@pytest.mark.parametrize("M,shape,expected", [
    # one sequence of 384 tokens (batch-1 denoising step)
    (384, QKV, (192, 128, 1, 0)), (384, O, (192, 128, 4, 0)), (384, GATEUP, (192, 256, 1, 0)), (384, DOWN, (192, 256, 8, 0)),
    # 192 tokens (BASELINE configuration 1): no padded tile row for the gate/up projection, QKV unsplit
    (192, QKV, (128, 128, 1, 0)), (192, GATEUP, (192, 128, 1, 0)), (192, DOWN, (192, 128, 8, 0)),
    # 32 rows (block-cached step, pruned last layer): weight-stream bound, 128-column tiles
    (32, QKV, (128, 128, 2, 0)), (32, DOWN, (128, 128, 8, 0)), (32, LM_HEAD, (128, 256, 1, 0)),
    # 8 x 384 tokens (the benchmark step): whole rounds of 192-row tiles, gate/up in the persistent form
    (3072, QKV, (192, 256, 1, 0)), (3072, O, (192, 256, 1, 0)), (3072, GATEUP, (192, 256, 1, 1)), (3072, DOWN, (192, 256, 1, 0)),
    # 8 x 640 and 2 x 1152 tokens: 256-row tiles where they fill the rounds better
    (5120, QKV, (256, 256, 1, 0)), (5120, GATEUP, (256, 256, 1, 0)), (5120, DOWN, (192, 256, 1, 0)), (2304, QKV, (256, 256, 1, 0)),
])
def test_plans_of_the_llada_8b_projections(lib, M, shape, expected):
    assert plan(lib, M, shape) == expected


def test_no_workspace_means_no_split(lib):
    for M in (32, 192, 384):
        for shape in (QKV, O, GATEUP, DOWN):
            assert plan(lib, M, shape, workspace=0)[2] == 1


def test_split_partials_fit_the_workspace(lib):
    for M in (32, 96, 192, 384, 576):
        for shape in (QKV, O, DOWN):
            N = shape[0]
            for ws in (M * N * 4 * 2, M * N * 4 * 5, M * N * 4 * 16):
                assert plan(lib, M, shape, workspace=ws)[2] * M * N * 4 <= ws
