import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library (ctypes handle); fails loudly if it was not built."""
    from llada_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def model8b(lib):
    """The mask predictor at the full LLaDA-8B shapes with random weights (seed 0), shared by the full-size tests."""
    import torch

    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    m = LLaDAModel(LLaDAWeights.random(LLaDAConfig(), "cuda", seed=0))
    yield m
    del m
    torch.cuda.empty_cache()
