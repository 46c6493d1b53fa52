"""Long randomised soak of every fuzzer in tests/fuzz_util.py: python tools/fuzz_soak.py [first_seed] [n_seeds] [cases per seed]."""
import sys
import time

sys.path.insert(0, ".")
from tests.fuzz_util import build_model, run, run_attention, run_gemm, run_likelihood, run_rows  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 200
built = build_model()
t0 = time.time()
for seed in range(first, first + n_seeds):
    a = run(cases, seed, built=built)
    run(cases // 2, seed + 10000, built=built, cache="dual")
    run(cases // 4, seed + 20000, built=built, cache="prefix")
    b = run_likelihood(cases, seed, built=built)
    c = run_attention(cases // 2, seed)
    run_gemm(cases, seed)
    run_rows(2 * cases, seed)
    print(f"seed {seed}: generate forward max-rel {a:.2e}, likelihood {b:.1e}, attention {c:.2e}, GEMM exact, rows ok, cached generate ok "
          f"[{time.time() - t0:.0f} s]", flush=True)
print("soak ok")
