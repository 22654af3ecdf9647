"""Parity links that the whole-chain replay alone cannot see, and the full-size configs  (-m gpu).

* The covariance factor the DEVICE uses for the beta draw (reference inference_utils.py:41,45):
  the replay tier builds its innovations by inverting the library's own basis, so it pins the
  mean, rss and sigma2 but returns ``mean_dev + (beta_ref - mean_ref)`` whatever sqrt(d_j) the
  kernel computes.  Here unit innovations e_j go through the loop kernel itself and
  sum_j (beta_j - mean)(beta_j - mean)' must equal the oracle's inv(X'X/s2 + P + 1e-6 I).
* K = 256 (the widest supported design matrix) replayed against the oracle, the Gram kernel at
  K = 256 against numpy, and 20 oracle iterations at the full C4 / C5 sizes.
* BASELINE configs[4]'s posterior-predictive leg at its stated size: 10000 draws x 50000
  held-out points x 257 models (reference sampling_utils.py:57-82).
"""
import numpy as np
import pandas as pd
import pytest

from gpu_common import golden_case, gpu_ctx
from oracle import bmc_oracle as O
from pybmc_amd import coverage
from pybmc_amd.chains import posterior_summary
from pybmc_amd.synthetic import synth_problem

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def dense_problem(n, k, seed, dt=np.float64):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(np.float64) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    A = rng.standard_normal((k, k))
    C0 = A @ A.T / k + np.eye(k)            # a dense, well-conditioned prior covariance
    prior = (rng.standard_normal(k) * 0.1, C0 * 50.0, 1.0, 0.02)
    return y, X, prior


# --------------------------------------------------------------------------- covariance
def device_cov_at(ctx, y, X, prior, s2_star):
    """cov and mean of beta | sigma2 = s2_star as the LOOP KERNEL realises them.

    K + 1 chains of two iterations.  Iteration 1 draws with xi = 0 (so beta_1 is the conditional
    mean at the OLS sigma2, which the oracle gives) and its Gamma variate is chosen so that the
    kernel's sigma2_1 = (nu0 s20 + rss(beta_1)) / (2 g_1) equals s2_star; iteration 2 then draws
    with xi = e_j (chain j < K) or 0 (chain K): beta_2 = mean(s2_star) + M e_j.
    Returns (sum_j (M e_j)(M e_j)', mean_2, sigma_1 recorded by the device)."""
    k = X.shape[1]
    Xf, yf = np.asarray(X, float), np.asarray(y, float)
    st = O.chain_setup(yf, Xf, prior)
    m0, _ = O.conditional_moments(st, yf, Xf, st["sigma2_init"])
    g1 = (st["nu0"] * st["s20"] + O.residual_rss(yf, Xf, m0)) / (2.0 * s2_star)
    xi = np.zeros((k + 1, 2, k))
    xi[np.arange(k), 1, np.arange(k)] = 1.0
    g = np.empty((k + 1, 2))
    g[:, 0] = g1
    g[:, 1] = 1.0
    out, _ = ctx.gibbs_run(k + 1, 2, xi=xi, g=g)
    assert rel(out[:, 0, :k], np.repeat(m0[None], k + 1, 0)) < 1e-10     # xi = 0: the mean itself
    mean2 = out[k, 1, :k]
    D = out[:k, 1, :k] - mean2                 # row j = (M e_j)'
    return D.T @ D, mean2, out[k, 0, k]


@pytest.mark.parametrize("case", ["gibbs_dense64x8", "c2", "k256", "k130_lds"])
def test_device_covariance_factor(case):
    """W diag(d) W' as computed by the kernel's rsqrt path == inv(X'X/s2 + P + 1e-6 I) of the
    oracle (reference inference_utils.py:41), at three values of sigma2."""
    ctx = gpu_ctx()
    if case == "gibbs_dense64x8":
        _, y, X, prior = golden_case(case)
    elif case == "c2":
        p = synth_problem(10000, 33, 32, seed=0)
        y, X, prior = p["y"], p["X"], p["prior"]
    elif case == "k256":
        y, X, prior = dense_problem(2000, 256, 3)
    else:
        y, X, prior = dense_problem(700, 130, 4)
    ctx.set_problem(y, X)
    ctx.set_prior(*prior)
    Xf = np.asarray(X, float)
    st = O.chain_setup(y, Xf, prior)
    for s2 in (st["sigma2_init"] * 1.7, 0.37, 5.0e-4):
        cov_dev, mean_dev, sig1 = device_cov_at(ctx, y, X, prior, s2)
        assert abs(sig1 ** 2 - s2) < 1e-12 * s2
        mean_o, cov_o = O.conditional_moments(st, y, Xf, sig1 ** 2)
        assert rel(cov_dev, cov_o) < 1e-10, (case, s2)
        assert rel(mean_dev, mean_o) < 1e-10, (case, s2)
        # and it is symmetric positive definite like the reference's (numpy checks this at :45)
        assert np.linalg.eigvalsh(0.5 * (cov_dev + cov_dev.T)).min() > 0


# --------------------------------------------------------------------------- K = 256
def replay_against_oracle(ctx, y, X, prior, T, seeds=(5, 6)):
    Xf, yf = np.asarray(X, float), np.asarray(y, float)
    st = O.chain_setup(yf, Xf, prior)
    Z, G = O.reference_streams(seeds[0], seeds[1], T, X.shape[1], O.gamma_shape(st))
    ref, trace = O.gibbs_replay(yf, Xf, T, prior, Z, G, return_sigma2=True)
    W, lam, s2i = ctx.basis()
    assert abs(s2i - st["sigma2_init"]) <= 1e-12 * st["sigma2_init"]
    xi = O.innovations_in_basis(st, yf, Xf, ref, W, lam, trace)
    out, stats = ctx.gibbs_run(1, T, xi=xi[None], g=G[None])
    return out[0], ref, stats


def test_t1_gram_at_k256():
    """f64 MFMA Gram at the widest K (17 x 17 tiles of 16 columns) against numpy."""
    ctx = gpu_ctx()
    y, X, prior = dense_problem(50000, 256, 8)
    ctx.set_problem(y, np.asfortranarray(X))
    Xa = np.column_stack([X, y])
    assert rel(ctx.gram(), Xa.T @ Xa) < 1e-12


@pytest.mark.parametrize("n,k", [(3, 2), (1237, 5), (4097, 47), (3000, 63), (7000, 79), (6000, 95),
                                 (9000, 130), (5000, 143), (5000, 144), (8000, 160), (2000, 255)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_t1_gram_every_tile_class(n, k, dt):
    """The Gram kernel chooses accumulators per wave, tile sets and row parts from the tile count
    (kernels_setup.hip, gram_geometry): one shape per class -- 6 / 8 / 10 / 20 accumulators, the
    row dimension split over waves or not, 64- and 32-row sub-panels, ragged row counts -- in f64
    and f32 storage against numpy (reference inference_utils.py:25,43: X.T.dot(X), X.T.dot(y))."""
    ctx = gpu_ctx()
    rng = np.random.Generator(np.random.PCG64(n + k))
    X = rng.standard_normal((n, k)).astype(dt)
    y = rng.standard_normal(n).astype(dt)
    ctx.set_problem(y, np.asfortranarray(X), dtype=dt)
    Xa = np.column_stack([X.astype(np.float64), y.astype(np.float64)])
    got = ctx.gram()
    assert rel(got, Xa.T @ Xa) < 1e-12
    assert np.array_equal(got, got.T)


@pytest.mark.parametrize("res", [0, 3])
def test_t2_replay_k256(res):
    """The oracle chain at K = 256 (N = 2000, dense prior covariance) through the HIP loop,
    LDS-pinned (auto) and streamed."""
    ctx = gpu_ctx()
    y, X, prior = dense_problem(2000, 256, 3)
    ctx.set_problem(y, X)
    ctx.set_prior(*prior)
    ctx.set_tuning(residency=res)
    out, ref, stats = replay_against_oracle(ctx, y, X, prior, 150)
    ctx.set_tuning()
    assert stats["residency"] == (3 if res == 3 else stats["residency"])
    assert np.abs(out - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    a, b = posterior_summary(out), posterior_summary(ref)
    for key in b:
        assert rel(a[key], b[key]) < 1e-6, key


def test_t2_replay_full_c5():
    """BASELINE configs[4] at full size (N = 50000, K = 256, float64): 20 iterations of the
    oracle replayed through the streaming loop."""
    ctx = gpu_ctx()
    y, X, prior = dense_problem(50000, 256, 8)
    ctx.set_problem(y, np.asfortranarray(X))
    ctx.set_prior(*prior)
    out, ref, stats = replay_against_oracle(ctx, y, X, prior, 20)
    assert stats["residency"] == 3
    assert np.abs(out - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    a, b = posterior_summary(out), posterior_summary(ref)
    for key in b:
        assert rel(a[key], b[key]) < 1e-6, key


def test_t2_replay_full_c4_float32():
    """BASELINE configs[3] at full size (N = 200000, K = 64, float32 storage, panels in VGPRs
    across the whole chip): 20 oracle iterations on the float32-rounded inputs.  Tolerance is
    the storage rounding (the rotated panels are rounded to float32 once more), 1e-5."""
    ctx = gpu_ctx()
    y, X, prior = dense_problem(200000, 64, 9, dt=np.float32)
    ctx.set_problem(y, np.asfortranarray(X), dtype=np.float32)
    ctx.set_prior(*prior)
    out, ref, stats = replay_against_oracle(ctx, y, X, prior, 20)
    assert stats["residency"] == 1
    assert np.abs(out - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    a, b = posterior_summary(out), posterior_summary(ref)
    for key in b:
        assert rel(a[key], b[key]) < 1e-5, key


# --------------------------------------------------------------------------- C5 predictive
def test_c5_predictive_full_size():
    """10000 draws x 50000 held-out points x 257 models (reference sampling_utils.py:57-82).

    (1) a 64-point slice with the reference's streams replayed == oracle.predictive_replay;
    (2) all points, sigma = 0: the f64 MFMA GEMM == numpy's W preds' (every one of 5e8 entries);
    (3) all points, device generator: the bands of a strided subset are numpy's percentiles of
        the returned draws exactly, the fused coverage counts over ALL points equal the host
        coverage() on the returned draws, the want_draws=False route (what evaluate() uses) gives
        the same bands and counts, and draws minus GEMM is standard-normal noise times sigma."""
    ctx = gpu_ctx()
    rng = np.random.Generator(np.random.PCG64(55))
    M, Km, k, S = 50000, 257, 256, 10000
    preds = rng.standard_normal((M, Km))                       # F_test = N(0,1)
    Vt_hat = rng.standard_normal((k, Km)) * 0.05
    samples = np.column_stack([rng.standard_normal((12000, k)) * 0.1,
                               rng.uniform(0.05, 0.15, 12000)])
    truth = preds.mean(1) + 0.1 * rng.standard_normal(M)
    pct = np.arange(0, 101, 5)
    q = (2.5, 50, 97.5)

    # (1) replay slice
    sl = slice(1000, 1064)
    r1 = np.random.Generator(np.random.PCG64(77))
    theta = r1.choice(samples, S, replace=False)                # :57
    noise = r1.standard_normal((S, 64))                         # :76
    ref_m, ref_bands = O.predictive_replay(preds[sl], samples, Vt_hat,
                                           np.random.Generator(np.random.PCG64(77)))
    got_m, got_bands, _ = ctx.predict(preds[sl], theta, Vt_hat, noise=noise, q=q)
    scale = np.abs(ref_m).max()
    assert np.abs(got_m - ref_m).max() < 1e-12 * scale
    assert np.abs(got_bands - np.array(ref_bands)).max() < 1e-12 * scale

    # (2) noiseless GEMM at full size
    theta0 = theta.copy()
    theta0[:, -1] = 0.0
    gemm, _, _ = ctx.predict(preds, theta0, Vt_hat, seed=9, q=())
    W = theta[:, :-1] @ Vt_hat + 1.0 / Km
    want = W @ preds.T
    assert gemm.shape == (S, M)
    assert np.abs(gemm - want).max() < 1e-12 * np.abs(want).max()
    del want

    # (3) device generator at full size
    draws, bands, cov = ctx.predict(preds, theta, Vt_hat, seed=9, q=q, truth=truth,
                                    cov_percentiles=pct)
    sub = slice(None, None, 97)
    assert np.array_equal(bands[:, sub], np.percentile(draws[:, sub], q, axis=0))
    assert coverage(pct, draws, pd.DataFrame({"truth": truth}), "truth") == cov
    _, bands2, cov2 = ctx.predict(preds, theta, Vt_hat, seed=9, q=q, truth=truth,
                                  cov_percentiles=pct, want_draws=False)
    assert np.array_equal(bands2, bands) and cov2 == cov
    assert cov[0] == 0.0 and all(b >= a for a, b in zip(cov, cov[1:])) and cov[-1] > 99
    z = (draws[:, sub] - gemm[:, sub]) / theta[:, -1][:, None]
    nz = z.size
    assert abs(z.mean()) < 5 / np.sqrt(nz) and abs(z.var() - 1) < 5 * np.sqrt(2 / nz)
    assert np.abs(np.corrcoef(z[:, :40].T)[~np.eye(40, dtype=bool)]).max() < 6 / np.sqrt(S)
