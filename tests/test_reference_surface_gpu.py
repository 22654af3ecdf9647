"""The reference's own tests, restated against this package's API  (-m gpu).
Sources: reference tests/test_inference_utils.py:5-19 and tests/test_bmc.py:83-120."""
import numpy as np
import pandas as pd
import pytest

from pybmc_amd import BayesianModelCombination, gibbs_sampler

pytestmark = pytest.mark.gpu


def test_gibbs_sampler_tiny():  # reference tests/test_inference_utils.py:5-19
    y = np.array([1.0, 2.0, 3.0])
    X = np.array([[1, 0], [0, 1], [1, 1]])
    prior_info = (np.array([0.0, 0.0]), np.eye(2), 1.0, 1.0)
    samples = gibbs_sampler(y, X, 10, prior_info)
    assert samples.shape == (10, 3)
    assert not np.any(np.isnan(samples))


def test_gibbs_sampler_seeded_by_global_numpy_stream():
    y = np.array([1.0, 2.0, 3.0])
    X = np.array([[1, 0], [0, 1], [1, 1]])
    prior_info = (np.array([0.0, 0.0]), np.eye(2), 1.0, 1.0)
    np.random.seed(5)
    a = gibbs_sampler(y, X, 50, prior_info)
    np.random.seed(5)
    b = gibbs_sampler(y, X, 50, prior_info)
    c = gibbs_sampler(y, X, 50, prior_info)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def make_bmc():
    df = pd.DataFrame({
        "x": [1, 2, 3, 4, 5, 6], "y": [10, 11, 12, 13, 14, 15],
        "truth": [11, 21, 31, 41, 51, 61], "model1": [10, 20, 30, 40, 50, 60],
        "model2": [15, 25, 35, 45, 55, 65], "model3": [12, 30, 32, 43, 58, 67]})
    return BayesianModelCombination(["model1", "model2", "model3", "truth"], {"target": df},
                                    "truth"), df


def test_train_default_options():  # reference tests/test_bmc.py:83-90 (50 000 iterations)
    bmc, df = make_bmc()
    bmc.orthogonalize("target", df.iloc[:4], 2)
    bmc.train()
    assert bmc.samples.shape == (50000, 3)
    assert not np.isnan(bmc.samples).any()
    # any sampler string other than "simplex" is Gibbs (quirk Q6)
    bmc.train({"iterations": 100, "sampler": "Gibbs_sampling"})
    assert bmc.samples.shape == (100, 3)


def test_train_two_row_frame():  # reference tests/test_bmc.py:92-105
    data = {"property": pd.DataFrame({"model1": [1, 2], "model2": [3, 4], "truth": [5, 6]})}
    bmc = BayesianModelCombination(["model1", "model2"], data, "truth")
    train_df = pd.DataFrame({"model1": [1, 2], "model2": [3, 4], "truth": [5, 6]})
    bmc.orthogonalize("property", train_df, 1)
    bmc.train({"iterations": 2000})
    assert bmc.samples is not None and bmc.samples.shape == (2000, 2)
    assert np.isfinite(bmc.samples).all()


def test_train_multi_chain_pools_samples():
    bmc, df = make_bmc()
    bmc.orthogonalize("target", df.iloc[:4], 2)
    bmc.train({"iterations": 300, "n_chains": 4, "seeds": [1, 2, 3, 4]})
    assert bmc.samples.shape == (1200, 3)
