"""Host-side mirror of the reference's ``pybmc/sampling_utils.py`` surface."""
from __future__ import annotations

import numpy as np

from . import _lib

N_PREDICTIVE_DRAWS = 10000  # reference sampling_utils.py:57


def rndm_m_random_calculator(filtered_model_predictions, samples, Vt_hat, *, seed=None,
                             device=0):
    """Posterior-predictive draws and 2.5/50/97.5 % bands on the GPU
    (reference sampling_utils.py:40-84).

    Returns ``(rndm_m, [lower, median, upper])`` with ``rndm_m`` of shape
    ``(10000, n_points)`` (C-ordered, like the reference's: sampling_utils.py:77).  Needs at least 10000 posterior samples, like the
    reference (``ValueError`` otherwise, :57).  The reference's side effect of
    re-seeding numpy's global stream (:54, quirk Q2) is not reproduced.
    """
    preds = np.ascontiguousarray(filtered_model_predictions, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    Vt_hat = np.ascontiguousarray(Vt_hat, dtype=np.float64)
    if samples.shape[0] < N_PREDICTIVE_DRAWS:
        raise ValueError("Cannot take a larger sample than population when replace is False")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 32, dtype=np.uint64)) << 32 | int(
            np.random.randint(0, 2 ** 32, dtype=np.uint64))
    ctx = _lib.default_context(device)
    rng = np.random.Generator(np.random.PCG64(seed))
    theta = rng.choice(samples, N_PREDICTIVE_DRAWS, replace=False)     # :57
    with ctx.lock:
        rndm_m, bands, _ = ctx.predict(preds, theta, Vt_hat, seed=seed)
    return rndm_m, [bands[0], bands[1], bands[2]]


def predictive_coverage(percentiles, filtered_model_predictions, samples, Vt_hat, truth, *,
                        seed=None, device=0):
    """``coverage(percentiles, rndm_m_random_calculator(...)[0], ...)`` fused on the GPU:
    the draws are sorted where they were produced and only the hit counts come back
    (what ``BayesianModelCombination.evaluate`` needs, reference bmc.py:366-376)."""
    preds = np.ascontiguousarray(filtered_model_predictions, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    if samples.shape[0] < N_PREDICTIVE_DRAWS:
        raise ValueError("Cannot take a larger sample than population when replace is False")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 32, dtype=np.uint64)) << 32 | int(
            np.random.randint(0, 2 ** 32, dtype=np.uint64))
    rng = np.random.Generator(np.random.PCG64(seed))
    theta = rng.choice(samples, N_PREDICTIVE_DRAWS, replace=False)
    ctx = _lib.default_context(device)
    with ctx.lock:
        _, _, cov = ctx.predict(preds, theta, Vt_hat, seed=seed, q=(), truth=truth,
                                cov_percentiles=list(percentiles), want_draws=False)
    return cov


def coverage(percentiles, rndm_m, models_output, truth_column):
    """Share of points whose truth lies inside the central p % credible interval,
    for each p (reference sampling_utils.py:4-37).  Index arithmetic (truncation
    towards zero, p = 0 never covers) is the reference's; each column is sorted
    once instead of once per percentile."""
    rndm_m = np.asarray(rndm_m)
    n_draws, n_points = rndm_m.shape
    truth = np.asarray(models_output[truth_column].tolist())
    srt = np.sort(rndm_m, axis=0)
    res = []
    for p in percentiles:
        lo = int((0.5 - p / 200) * n_draws)
        hi = int((0.5 + p / 200) * n_draws) - 1
        inside = (srt[lo] <= truth) & (truth <= srt[hi])
        res.append(int(np.count_nonzero(inside)) / n_points * 100)
    return res
