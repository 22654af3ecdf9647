"""pybmc_amd: MI355X-native Gibbs-sampling core for Bayesian model combination.

Public names follow the reference package (pybmc/__init__.py:11-24); the
nonexistent ``Model`` of its ``__all__`` is dropped.
"""
from .bmc import BayesianModelCombination
from .data import Dataset
from .inference_utils import gibbs_sampler, gibbs_sampler_simplex, USVt_hat_extraction
from .sampling_utils import coverage, rndm_m_random_calculator

__all__ = [
    "Dataset",
    "BayesianModelCombination",
    "gibbs_sampler",
    "gibbs_sampler_simplex",
    "USVt_hat_extraction",
    "coverage",
    "rndm_m_random_calculator",
]
