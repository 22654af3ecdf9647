"""pybmc_amd: MI355X-native Gibbs-sampling core for Bayesian model combination.

Public names follow the reference package (pybmc/__init__.py:11-24).
"""
from .inference_utils import gibbs_sampler, USVt_hat_extraction

__all__ = ["gibbs_sampler", "USVt_hat_extraction"]
