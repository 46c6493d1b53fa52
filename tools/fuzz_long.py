"""A longer slice of the randomised checks of tests/fuzz_util.py than the test suite runs (new seeds), one pass:
python tools/fuzz_long.py [first_seed] [seeds]"""
import sys
sys.path.insert(0, ".")
from tests import fuzz_util as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for seed in range(first, first + n):
    g = F.run_gemm(60, seed)
    a = F.run_attention(30, seed)
    r = F.run_rows(60, seed)
    gen = F.run(20, seed)
    ll = F.run_likelihood(15, seed)
    print(f"seed {seed}: gemm {g}/60 exact, attention max err {a:.2e}, remask rows {r}/60, generate max deviation {gen:.2e}, likelihood rel err {ll:.2e}", flush=True)
    assert g == 60 and r == 60 and a <= 1.5e-2 and gen <= 2e-2 and ll <= 1e-3
print("all good")
