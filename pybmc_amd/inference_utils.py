"""Host-side mirror of the reference's ``pybmc/inference_utils.py`` surface.

``gibbs_sampler`` keeps the reference signature and return layout
(reference inference_utils.py:4-56) and runs on the MI355X through the C ABI.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _draw_seeds(n):
    """Seeds come from numpy's legacy global stream -- the stream the reference's
    beta draw consumes (inference_utils.py:45) -- so ``np.random.seed(s)`` before a
    call makes a run repeatable (the reference itself is not: its sigma2 draw uses
    an unseeded generator, inference_utils.py:52)."""
    hi = np.random.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    lo = np.random.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    return (hi << np.uint64(32)) | lo


def gibbs_sampler(y, X, iterations, prior_info, *, n_chains=1, seeds=None, device=0,
                  dtype=None, return_stats=False, rss="data", _problem_on_device=False):
    """Gibbs sampling for Bayesian linear regression on the GPU.

    Same arguments and result as the reference (inference_utils.py:4-20):
    ``prior_info = (b_mean_prior, b_mean_cov, nu0, sigma20)``; returns an
    ``(iterations, k+1)`` array whose rows are ``[beta, sigma]`` (the last column is
    sigma, not sigma**2, :54).  Raises ``numpy.linalg.LinAlgError`` where the
    reference's ``inv`` calls do (:22, :26).

    Extensions (keyword-only, defaults preserve the reference behaviour):
    ``n_chains`` > 1 returns ``(n_chains, iterations, k+1)``; ``seeds`` fixes the
    per-chain Philox keys; ``dtype=np.float32`` stores X and y in float32 (sums
    stay float64); ``rss="gram"`` (opt-in, at most 64 columns) takes the residual sum of
    squares of :48-51 from sufficient statistics instead of a pass over the data in every
    iteration -- the same chain up to rounding of that sum, one wave per chain.
    """
    if rss not in ("data", "gram"):
        raise ValueError('rss must be "data" or "gram"')
    b0, C0, nu0, s20 = prior_info
    ctx = _lib.default_context(device)
    with ctx.lock:   # (the per-device context is shared: one caller's sequence at a time)
        if not _problem_on_device:   # orthogonalize(method="device") left (y, X) on the GPU
            ctx.set_problem(y, X, dtype=dtype)
        ctx.set_prior(b0, C0, nu0, s20)
        if seeds is None:
            seeds = _draw_seeds(n_chains)
        if rss == "gram":
            ctx.set_tuning(rss_mode=1)
        try:
            out, stats = ctx.gibbs_run(n_chains, int(iterations), seeds=seeds)
        finally:
            if rss == "gram":
                ctx.set_tuning()
    res = out[0] if n_chains == 1 else out
    return (res, stats) if return_stats else res


def USVt_hat_extraction(U, S, Vt, components_kept):
    """Truncate an SVD to ``components_kept`` components
    (reference inference_utils.py:147-168).  ``U_hat`` is returned column-major
    (F-contiguous), which is also the layout the device kernels read natively."""
    k = int(components_kept)
    U = np.asarray(U)
    S = np.asarray(S)
    Vt = np.asarray(Vt)
    U_hat = np.asfortranarray(U[:, :k])
    S_hat = S[:k]
    Vt_hat_normalized = np.array(Vt[:k])
    Vt_hat = Vt_hat_normalized / S_hat[:, None]
    return U_hat, S_hat, Vt_hat, Vt_hat_normalized


def gibbs_sampler_simplex(y, X, Vt_hat, S_hat, iterations, prior_info, burn=10000,
                          stepsize=0.001, *, seed=None, device=0):
    """Random-walk Metropolis on the weight simplex with a Gibbs sigma2 step
    (reference inference_utils.py:59-144).  Same arguments, result, ``ValueError``s
    (:91-94) and acceptance-rate print (:143) as the reference."""
    if burn < 0:
        raise ValueError("Burn-in iterations must be non-negative.")
    if stepsize <= 0:
        raise ValueError("Stepsize must be positive.")
    nu0, s20 = prior_info
    ctx = _lib.default_context(device)
    with ctx.lock:
        ctx.set_problem(y, X)
        if seed is None:
            seed = int(_draw_seeds(1)[0])
        samples, accepted = ctx.simplex_run(Vt_hat, S_hat, int(iterations), float(nu0), float(s20),
                                            int(burn), float(stepsize), seed=seed)
    print(f"Acceptance rate: {accepted / iterations * 100:.2f}%")
    return samples
