"""Requests per second of the likelihood estimator on LLaDA-8B shapes for single-row requests (mc_num =
batch_size = 1, as the reference runs MMLU-style tasks, eval_llada_lm_eval.sh:12-28): one-by-one calls against
get_log_likelihood_many sharing right-padded forwards."""
import sys
import time

import torch

sys.path.insert(0, ".")
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights, get_log_likelihood, get_log_likelihood_many  # noqa: E402

model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda"))
g = torch.Generator().manual_seed(3)
n = 64
pairs = [(torch.randint(0, 126000, (int(torch.randint(300, 500, (1,), generator=g)),), generator=g),
          torch.randint(0, 126000, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)) for _ in range(n)]
seeds = list(range(n))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


t1, one = timed(lambda: [get_log_likelihood(model, p, a, mc_num=1, batch_size=1, seed=s) for (p, a), s in zip(pairs, seeds)])
print(f"one by one: {n / t1:7.1f} requests/s ({1e3 * t1 / n:.2f} ms each)", flush=True)
for rows in (8, 16, 32):
    t2, many = timed(lambda: get_log_likelihood_many(model, pairs, mc_num=1, batch_size=1, seeds=seeds, max_rows=rows))
    worst = max(abs(a - b) / abs(b) for a, b in zip(many, one))
    print(f"{rows:3d} rows per forward: {n / t2:7.1f} requests/s ({1e3 * t2 / n:.2f} ms each), worst relative difference {worst:.1e}",
          flush=True)
