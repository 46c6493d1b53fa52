"""Randomised end-to-end check of llada_amd.generate against the oracle on a GPU box (tests/fuzz_util.py):

    python tools/fuzz_generate.py [iterations] [seed]
"""
import sys

sys.path.insert(0, ".")
from tests.fuzz_util import build_model, run, run_attention, run_gemm, run_likelihood, run_rows  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
built = build_model()
worst = run(n_iter, seed, verbose=True, built=built)
print(f"fuzz ok: {n_iter} runs, worst forward max-rel {worst:.2e}")
print(f"likelihood fuzz ok: {n_iter} runs, worst relative difference {run_likelihood(n_iter, seed, built=built):.2e}")
print(f"attention fuzz ok: {n_iter} runs, worst error {run_attention(n_iter, seed):.2e} of the largest value")
print(f"GEMM fuzz ok: {run_gemm(n_iter, seed)} runs, all bit-exact")
print(f"row-kernel fuzz ok: {run_rows(n_iter, seed)} runs")
