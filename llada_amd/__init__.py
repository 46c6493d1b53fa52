"""llada_amd -- MI355X-native LLaDA sampling and likelihood.

Drop-in surface (same signatures as the reference's generate.py / get_log_likelihood.py):
    from llada_amd import generate, get_log_likelihood
plus the model object the hot path runs on:
    from llada_amd import LLaDAModel, LLaDAWeights, LLaDAConfig
"""
from .generate import add_gumbel_noise, generate, get_num_transfer_tokens   # noqa: F401
from .get_log_likelihood import forward_process, get_log_likelihood, get_log_likelihood_many   # noqa: F401
from .model import LLaDAModel, ModelOutput                                  # noqa: F401
from .weights import LLaDAConfig, LLaDAWeights                              # noqa: F401
