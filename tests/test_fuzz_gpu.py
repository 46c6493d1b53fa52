"""Fixed-seed slice of the randomised generate() check (tests/fuzz_util.py)."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_generate_random_cases(lib, seed):
    from tests.fuzz_util import run
    assert run(25, seed) <= 2e-2


@pytest.mark.parametrize("seed", [1, 2])
def test_likelihood_random_cases(lib, seed):
    from tests.fuzz_util import run_likelihood
    assert run_likelihood(25, seed) <= 1e-3


@pytest.mark.parametrize("seed", [1, 2])
def test_attention_random_cases(lib, seed):
    from tests.fuzz_util import run_attention
    assert run_attention(20, seed) <= 1.5e-2


@pytest.mark.parametrize("seed", [1, 2])
def test_gemm_random_cases(lib, seed):
    from tests.fuzz_util import run_gemm
    assert run_gemm(30, seed) == 30


@pytest.mark.parametrize("seed", [1, 2])
def test_remask_random_cases(lib, seed):
    from tests.fuzz_util import run_rows
    assert run_rows(40, seed) == 40


def test_generate_random_cases_block_cached(lib):
    from tests.fuzz_util import run
    assert run(25, 4, cache="dual") <= 2e-2
    assert run(15, 6, cache="prefix") <= 2e-2
