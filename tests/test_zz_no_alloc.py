"""llada_remask_scores_static (the entry point that allocates nothing: its piece statistics live in a device
array built into the library) against llada_remask_scores with a caller workspace, through the C ABI:
bit-identical candidates and confidences on two calls in a row, the second on a smaller input, for the
temperature-0 rules, the float64 Gumbel path (Philox noise) and classifier-free guidance; sizes beyond the
built-in scratch are refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MASK_ID, EOS_ID, EOT_ID = 126336, 126081, 126348


def _run(lib, entry, logits, uncond, cfg, x, block_end, rule, flags, T, seed, step, workspace):
    from llada_amd import _lib
    B, L = x.shape
    n_rows, V = logits.shape
    x0 = torch.full((B, L), -7, dtype=torch.long, device="cuda")
    conf = torch.full((B, L), 123.0, dtype=torch.float32, device="cuda")
    args = _lib.RemaskArgs(logits.data_ptr(), _lib.ptr(uncond), float(cfg), None, n_rows, V, x.data_ptr(), B, L, block_end,
                           MASK_ID, _lib.REMASK[rule], flags, EOS_ID, EOT_ID, float(T), None, None, seed, step, 0,
                           x0.data_ptr(), conf.data_ptr())
    if workspace is not None:
        args.workspace, args.workspace_bytes = workspace.data_ptr(), workspace.numel()
    rc = entry(C.byref(args), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.llada_last_error().decode()
    torch.cuda.synchronize()
    return x0.cpu().numpy(), conf.cpu().numpy()


def _inputs(seed, B, L, V, n_masked_from):
    g = torch.Generator(device="cuda").manual_seed(seed)
    logits = (torch.randn(B * L, V, device="cuda", generator=g) * 3).to(torch.bfloat16)
    uncond = (torch.randn(B * L, V, device="cuda", generator=g) * 3).to(torch.bfloat16)
    x = torch.randint(0, 1000, (B, L), device="cuda", generator=g)
    x[:, n_masked_from:] = MASK_ID
    return logits, uncond, x


@pytest.mark.parametrize("rule,T,cfg,flags", [("low_confidence", 0.0, 0.0, 0), ("high_confidence", 0.0, 1.5, 3),
                                              ("random", 0.0, 0.0, 0), ("low_confidence", 0.7, 0.0, 2),
                                              ("low_entropy", 0.0, 0.0, 0)])
def test_static_scratch_matches_workspace_call(lib, rule, T, cfg, flags):
    # two calls in a row on one device, the second on a smaller input (fewer rows AND a shorter vocabulary, so
    # stale statistics of the first call lie past everything the second one writes)
    for (B, L, V, first_masked) in [(3, 24, 126464, 5), (2, 9, 126400, 2)]:
        logits, uncond, x = _inputs(17 + B, B, L, V, first_masked)
        un = uncond if cfg > 0 else None
        block_end = L - 3
        ws = torch.empty(lib.llada_rows_workspace_bytes(B * L), dtype=torch.uint8, device="cuda")
        want = _run(lib, lib.llada_remask_scores, logits, un, cfg, x, block_end, rule, flags, T, 5, 2, ws)
        got = _run(lib, lib.llada_remask_scores_static, logits, un, cfg, x, block_end, rule, flags, T, 5, 2, None)
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        # something was actually scored
        assert np.isfinite(want[1]).sum() == B * (block_end - first_masked)


def test_static_scratch_rejects_rows_beyond_its_size(lib):
    from llada_amd import _lib
    n = _lib.STATIC_REMASK_MAX_ROWS + 1
    one = 16    # never dereferenced: the size check comes first
    args = _lib.RemaskArgs(one, None, 0.0, None, n, 8, one, 1, n, n, MASK_ID, 0, 0, EOS_ID, EOT_ID, 0.0, None, None, 0, 0, 0,
                           one, one)
    assert lib.llada_remask_scores_static(C.byref(args), None) == _lib.ERR_WORKSPACE
    assert b"8192" in lib.llada_last_error()
