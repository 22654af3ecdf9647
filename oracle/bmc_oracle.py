"""CPU oracle for the pyBMC Gibbs hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference algorithm.  It exists so
that the HIP product path can be *checked*; nothing under ``pybmc_amd/`` may
import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the unmodified
reference (in the build container only), pins its two random streams from the
outside and stores its outputs; ``tests/test_oracle_golden.py`` checks every
function below against those files bit-for-bit (max |diff| == 0.0).

Reference lines restated (paths relative to the reference checkout):

* ``chain_setup``            <- pybmc/inference_utils.py:21-37
* ``conditional_moments``    <- pybmc/inference_utils.py:41-44
* ``mvn_draw_svd``           <- pybmc/inference_utils.py:45  (numpy legacy
                                ``RandomState.multivariate_normal``: svd map)
* ``residual_rss``           <- pybmc/inference_utils.py:48-51
* ``sigma2_draw``            <- pybmc/inference_utils.py:50-52
* ``gibbs_replay``           <- pybmc/inference_utils.py:39-56 (explicit streams)
* ``gibbs_port``             <- pybmc/inference_utils.py:4-56  (same numpy calls;
                                this is the timed CPU baseline)
* ``simplex_replay``         <- pybmc/inference_utils.py:78-144
* ``usvt_hat``               <- pybmc/inference_utils.py:147-168
* ``centre_and_svd``         <- pybmc/bmc.py:102-122
* ``predictive_replay``      <- pybmc/sampling_utils.py:54-84
* ``coverage_oracle``        <- pybmc/sampling_utils.py:18-37

How the reference consumes randomness (needed to replay it):

* beta draw: ``np.random.multivariate_normal`` on the *legacy global*
  ``RandomState``: one row of K standard normals per iteration, so the whole
  stream is ``RandomState(seed_z).standard_normal((T, K))``.
* sigma2 draw: ``np.random.default_rng().gamma(a, s)`` equals
  ``s * standard_gamma(a)`` bit-for-bit; with one persistent generator the
  stream is ``Generator(PCG64(seed_g)).standard_gamma(a, size=T)``.
"""
from __future__ import annotations

import numpy as np

RIDGE = 1e-6        # inference_utils.py:41
SIGMA2_FLOOR = 1e-6  # inference_utils.py:37,52
N_PREDICTIVE = 10000  # sampling_utils.py:57


# --------------------------------------------------------------------------
# gibbs_sampler
# --------------------------------------------------------------------------
def chain_setup(y, X, prior_info):
    """One-off quantities of inference_utils.py:21-37.

    Returns a dict with P (prior precision), XtX, n, sigma2_init, plus the raw
    prior entries.  The dead store at :32 is not restated.
    """
    b0, C0, nu0, s20 = prior_info
    P = np.linalg.inv(C0)                       # :22
    n = len(y)                                  # :23
    XtX = X.T.dot(X)                            # :25
    XtX_inv = np.linalg.inv(XtX)                # :26
    b_ols = XtX_inv.dot(X.T).dot(y)             # :28
    res = y - X.dot(b_ols)                      # :29-30
    s2 = np.sum(res ** 2) / len(res)            # :31
    s2 = max(s2, SIGMA2_FLOOR)                  # :37
    return dict(P=P, XtX=XtX, n=n, sigma2_init=s2, b0=b0, nu0=nu0, s20=s20)


def conditional_moments(st, y, X, sigma2):
    """cov, mean of beta | sigma2  (inference_utils.py:41-44)."""
    XtX, P = st["XtX"], st["P"]
    cov = np.linalg.inv(XtX / sigma2 + P + np.eye(XtX.shape[0]) * RIDGE)
    mean = cov.dot(P.dot(st["b0"]) + X.T.dot(y) / sigma2)
    return mean, cov


def mvn_draw_svd(mean, cov, z):
    """beta = mean + (z * sqrt(s)) @ v with (u, s, v) = svd(cov).

    This is what numpy's legacy ``multivariate_normal`` does with its row of
    standard normals ``z`` (inference_utils.py:45).  The (1,K)x(K,K) ``dot`` is
    kept in that shape so the BLAS call, hence the rounding, is the same.
    """
    _, s, v = np.linalg.svd(cov)
    x = np.dot(z.reshape(1, -1), np.sqrt(s)[:, None] * v)
    x += mean
    return x.reshape(-1)


def residual_rss(y, X, beta):
    """rss = sum((y - X beta)^2)  (inference_utils.py:48-51)."""
    r = y - X.dot(beta)
    return np.sum(r ** 2)


def sigma2_draw(st, rss, g):
    """sigma2 from a standard gamma variate g (inference_utils.py:50-52).

    ``Generator.gamma(shape, scale)`` is ``scale * standard_gamma(shape)``.
    """
    scale_post = (st["nu0"] * st["s20"] + rss) / 2.0
    return max(1 / (g * (1 / scale_post)), SIGMA2_FLOOR)


def gamma_shape(st):
    return (st["nu0"] + st["n"]) / 2.0           # :50


def gibbs_replay(y, X, iterations, prior_info, Z, G, return_sigma2=False):
    """The reference chain driven by explicit streams Z (T,K) and G (T,).

    With ``return_sigma2`` also returns the exact sigma2 trace (T+1,), entry 0
    being the OLS initial value, so callers need not square the stored sigma.
    """
    st = chain_setup(y, X, prior_info)
    s2 = st["sigma2_init"]
    K = X.shape[1]
    out = np.empty((iterations, K + 1))
    trace = np.empty(iterations + 1)
    trace[0] = s2
    for t in range(iterations):
        mean, cov = conditional_moments(st, y, X, s2)
        beta = mvn_draw_svd(mean, cov, Z[t])
        s2 = sigma2_draw(st, residual_rss(y, X, beta), G[t])
        out[t, :K] = beta
        out[t, K] = np.sqrt(s2)                  # :54 (sigma, not sigma2)
        trace[t + 1] = s2
    return (out, trace) if return_sigma2 else out


def reference_streams(seed_z, seed_g, iterations, K, shape):
    """The variates the pinned reference consumes (see module docstring)."""
    Z = np.random.RandomState(seed_z).standard_normal((iterations, K))
    G = np.random.Generator(np.random.PCG64(seed_g)).standard_gamma(
        shape, size=iterations)
    return Z, G


def gibbs_port(y, X, iterations, prior_info):
    """Same numpy calls per iteration as inference_utils.py:39-54, including the
    loop-invariant X.T.dot(y), the SVD-based draw and a fresh default_rng() per
    iteration.  Unseeded like the reference; used as the timed CPU baseline."""
    st = chain_setup(y, X, prior_info)
    s2 = st["sigma2_init"]
    XtX, P, b0 = st["XtX"], st["P"], st["b0"]
    n, nu0, s20 = st["n"], st["nu0"], st["s20"]
    rows = []
    for _ in range(iterations):
        cov = np.linalg.inv(XtX / s2 + P + np.eye(XtX.shape[0]) * RIDGE)
        mean = cov.dot(P.dot(b0) + X.T.dot(y) / s2)
        beta = np.random.multivariate_normal(mean, cov)
        r = y - X.dot(beta)
        a = (nu0 + n) / 2.0
        sc = (nu0 * s20 + np.sum(r ** 2)) / 2.0
        s2 = max(1 / np.random.default_rng().gamma(a, 1 / sc), SIGMA2_FLOOR)
        rows.append(np.append(beta, np.sqrt(s2)))
    return np.array(rows)


# --------------------------------------------------------------------------
# gibbs_sampler_simplex
# --------------------------------------------------------------------------
def simplex_replay(y, X, Vt_hat, S_hat, iterations, prior_info, burn, stepsize,
                   Z, U, G, means_out=None):
    """inference_utils.py:78-144 with explicit streams.

    Z: (burn+iterations, K) standard normals of the proposal draw (:98,:121),
    U: uniforms, consumed ONLY when the proposal is inside the simplex
       (:110,:132) -> U is indexed by a running counter,
    G: (burn+iterations,) standard gamma variates (:117,:140).
    Returns (samples, acceptance_count, uniforms_used).
    """
    nm = Vt_hat.shape[1]
    bias0 = np.full(nm, 1 / nm)                                  # :78
    nu0, s20 = prior_info
    step_cov = np.diag(S_hat ** 2 * stepsize ** 2)               # :80
    n = len(y)
    b_cur = np.full(X.shape[1], 0)                               # :82
    ll_cur = -np.sum((y - X.dot(b_cur)) ** 2)                    # :83-85
    s2 = -ll_cur / n                                             # :86
    if burn < 0:
        raise ValueError("Burn-in iterations must be non-negative.")
    if stepsize <= 0:
        raise ValueError("Stepsize must be positive.")
    out, acc, iu = [], 0, 0
    for t in range(burn + iterations):
        if means_out is not None:
            means_out.append(np.asarray(b_cur, dtype=float).copy())
        b_prop = mvn_draw_svd(np.asarray(b_cur, dtype=float), step_cov, Z[t])
        omegas = np.dot(b_prop, Vt_hat) + bias0                  # :99,:122
        if not np.any(omegas < 0):
            ll_prop = -np.sum((y - X.dot(b_prop)) ** 2)
            p_acc = min(1, np.exp((ll_prop - ll_cur) / s2))
            u = U[iu]
            iu += 1
            if u < p_acc:
                b_cur = np.copy(b_prop)
                ll_cur = ll_prop
                if t >= burn:
                    acc += 1
        a = (nu0 + n) / 2.0
        sc = (nu0 * s20 - ll_cur) / 2.0
        s2 = 1 / (G[t] * (1 / sc))                               # no floor here
        if t >= burn:
            out.append(np.append(b_cur, np.sqrt(s2)))
    return np.array(out), acc, iu


# --------------------------------------------------------------------------
# orthogonalize / USVt
# --------------------------------------------------------------------------
def usvt_hat(U, S, Vt, k):
    """inference_utils.py:164-167.  U_hat comes out F-contiguous."""
    U_hat = np.array([U.T[i] for i in range(k)]).T
    S_hat = S[:k]
    Vt_hat = np.array([Vt[i] / S[i] for i in range(k)])
    Vt_norm = np.array([Vt[i] for i in range(k)])
    return U_hat, S_hat, Vt_hat, Vt_norm


def centre_and_svd(F, truth, k, full_matrices=True):
    """bmc.py:106-122: row-mean over models, centring, SVD, truncation."""
    mu = np.mean(F, axis=1)
    yc = truth - mu
    Fc = F - mu[:, None]
    U, S, Vt = np.linalg.svd(Fc, full_matrices=full_matrices)
    return (mu, yc) + usvt_hat(U, S, Vt, k)


# --------------------------------------------------------------------------
# posterior predictive + coverage
# --------------------------------------------------------------------------
def predictive_replay(preds, samples, Vt_hat, rng):
    """sampling_utils.py:57-82 driven by an explicit Generator ``rng``."""
    theta = rng.choice(samples, N_PREDICTIVE, replace=False)
    betas, sig = theta[:, :-1], theta[:, -1]
    nm = Vt_hat.shape[1]
    W = betas @ Vt_hat + np.full(nm, 1 / nm)
    Y = W @ preds.T
    rndm_m = Y + rng.standard_normal(Y.shape) * sig[:, None]
    bands = [np.percentile(rndm_m, q, axis=0) for q in (2.5, 50, 97.5)]
    return rndm_m, bands


def coverage_oracle(percentiles, rndm_m, truth):
    """sampling_utils.py:18-37, one sort per column instead of 21."""
    M = len(rndm_m)
    srt = np.sort(rndm_m, axis=0)
    truth = np.asarray(truth)
    res = []
    for p in percentiles:
        lo = int((0.5 - p / 200) * M)
        hi = int((0.5 + p / 200) * M) - 1
        hit = (srt[lo] <= truth) & (truth <= srt[hi])
        res.append(np.count_nonzero(hit) / rndm_m.shape[1] * 100)
    return res


# --------------------------------------------------------------------------
# helpers shared by the parity tests (not part of the reference)
# --------------------------------------------------------------------------
def innovations_in_basis(st, y, X, samples, W, lam, sigma2_trace=None):
    """Express the oracle's beta draws as innovations xi in the product's basis.

    The product draws beta = W (d*(c1 + c2/s2) + sqrt(d)*xi) with
    d = 1/(lam/s2 + 1)  (see DESIGN.md "rotated draw").  Given the oracle's
    chain, xi_t = diag(1/sqrt(d_t)) W^{-1} (beta_t - mean_t).
    """
    T, K1 = samples.shape
    K = K1 - 1
    Winv = np.linalg.inv(W)
    xi = np.empty((T, K))
    s2 = st["sigma2_init"]
    for t in range(T):
        mean, _ = conditional_moments(st, y, X, s2)
        d = 1.0 / (lam / s2 + 1.0)
        xi[t] = Winv.dot(samples[t, :K] - mean) / np.sqrt(d)
        s2 = samples[t, K] ** 2 if sigma2_trace is None else sigma2_trace[t + 1]
    return xi
