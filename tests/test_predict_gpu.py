"""Posterior predictive on the GPU (-m gpu): reference sampling_utils.py:40-84 and the
interval test of coverage (:24-34), through the C ABI.

Replay tier: the reference's own run (golden fixture: selection and noise streams pinned from
outside) is reproduced with its selected rows and its noise; tolerance 1e-12 relative on the
draws and bands, equality on the coverage percentages.  Device-generator tier: distributional.
"""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from gpu_common import gpu_ctx
from oracle import bmc_oracle as O
from pybmc_amd import BayesianModelCombination, coverage, rndm_m_random_calculator
from pybmc_amd._lib import coverage_plan, order_stat_plan

pytestmark = pytest.mark.gpu


def test_order_stat_plan_matches_numpy():
    rng = np.random.default_rng(0)
    for n in (10000, 9999, 64, 3):
        x = np.sort(rng.standard_normal(n))
        qi, qg = order_stat_plan(n, (2.5, 50, 97.5, 0, 100))
        for (i, g), p in zip(zip(qi, qg), (2.5, 50, 97.5, 0, 100)):
            a, b = x[i], x[min(i + 1, n - 1)]
            d = b - a
            got = b - d * (1 - g) if g >= 0.5 else a + d * g
            assert got == np.percentile(x, p)


def test_replay_matches_the_reference():
    g = load_golden("predict_synth48")
    ctx = gpu_ctx()
    rng = np.random.Generator(np.random.PCG64(int(g["seed_g"])))
    theta = rng.choice(g["samples"], 10000, replace=False)       # sampling_utils.py:57
    noise = rng.standard_normal((10000, g["preds"].shape[0]))    # :76
    rng2 = np.random.Generator(np.random.PCG64(int(g["seed_g"])))
    ref_m, ref_bands = O.predictive_replay(g["preds"], g["samples"], g["Vt_hat"], rng2)
    assert np.array_equal(ref_m[:64], g["rndm_m_head"])
    pct = np.arange(0, 101, 5)
    rndm_m, bands, cov = ctx.predict(g["preds"], theta, g["Vt_hat"], noise=noise, truth=g["truth"],
                                     cov_percentiles=pct)
    assert rndm_m.shape == (10000, 48)
    scale = np.abs(ref_m).max()
    assert np.abs(rndm_m - ref_m).max() < 1e-12 * scale
    for got, want in zip(bands, (g["lower"], g["median"], g["upper"])):
        assert np.abs(got - want).max() < 1e-12 * scale
    assert cov == list(g["coverage"])
    # the host-side coverage() on the returned draws agrees with the fused device count
    df = pd.DataFrame({"truth": g["truth"]})
    assert coverage(pct, rndm_m, df, "truth") == cov


@pytest.mark.parametrize("M,Km,k,S", [(1, 2, 1, 64), (65, 5, 3, 1000), (130, 33, 32, 10000),
                                      (7, 257, 9, 4097)])
def test_replay_shapes(M, Km, k, S):
    ctx = gpu_ctx()
    rng = np.random.default_rng(M + Km)
    preds = rng.standard_normal((M, Km)) + 3
    theta = np.column_stack([rng.standard_normal((S, k)) * 0.1, rng.uniform(0.5, 1.5, S)])
    Vt = rng.standard_normal((k, Km))
    noise = rng.standard_normal((S, M))
    W = theta[:, :-1] @ Vt + 1.0 / Km
    ref = W @ preds.T + noise * theta[:, -1][:, None]
    q = (2.5, 50, 97.5)
    rndm_m, bands, _ = ctx.predict(preds, theta, Vt, noise=noise, q=q)
    assert np.abs(rndm_m - ref).max() < 1e-11 * np.abs(ref).max()
    want = np.percentile(ref, q, axis=0)
    assert np.abs(bands - want).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("kind", ["normal", "ties", "outlier", "flat", "two_level"])
@pytest.mark.parametrize("S", [2048, 10000, 16384])
def test_order_statistics_by_selection_are_exact(kind, S):
    """S >= 2048 draws: the requested ranks are found by histogram selection, and rows the
    selection cannot resolve (a requested rank inside a bin of many equal draws, a range
    stretched by an outlier) go through the full sort.  Either way the percentiles and coverage
    counts are those of numpy on the returned draws, exactly."""
    ctx = gpu_ctx()
    rng = np.random.default_rng(S + len(kind))
    M, Km, k = 37, 3, 2
    preds = rng.standard_normal((M, Km)) + 2
    theta = np.column_stack([np.zeros((S, k)), np.ones(S)])          # weights 1/Km, sigma 1
    Vt = rng.standard_normal((k, Km))
    noise = rng.standard_normal((S, M))
    if kind == "ties":
        noise[: S // 2] = 0.25                      # half of every point's draws are one value
    elif kind == "outlier":
        noise[3] = 1e9                              # one far draw per point stretches [min, max]
    elif kind == "flat":
        noise[:] = -1.5                             # every draw of a point equal
    elif kind == "two_level":
        noise[:] = rng.integers(0, 2, size=(S, M))  # only two distinct values per point
    truth = preds.mean(1) + rng.standard_normal(M)
    pct = np.arange(0, 101, 5)
    q = (2.5, 50, 97.5, 0, 100, 33.3)
    rndm_m, bands, cov = ctx.predict(preds, theta, Vt, noise=noise, q=q, truth=truth,
                                     cov_percentiles=pct)
    assert np.array_equal(bands, np.percentile(rndm_m, q, axis=0))
    df = pd.DataFrame({"truth": truth})
    assert coverage(pct, rndm_m, df, "truth") == cov


def test_device_generator_distribution():
    g = load_golden("predict_synth48")
    np.random.seed(3)
    rndm_m, (lo, med, up) = rndm_m_random_calculator(g["preds"], g["samples"], g["Vt_hat"])
    assert rndm_m.shape == (10000, 48) and np.isfinite(rndm_m).all()
    sd = rndm_m.std(0)
    # Monte-Carlo error of a quantile of 10000 draws: ~ sd * sqrt(q(1-q)/n) / pdf
    assert np.all(np.abs(med - g["median"]) < 0.08 * sd)
    assert np.all(np.abs(lo - g["lower"]) < 0.2 * sd)
    assert np.all(np.abs(up - g["upper"]) < 0.2 * sd)
    # the noise is standard normal and independent across points
    z = (rndm_m - rndm_m.mean(0)) / sd
    c = np.corrcoef(z.T)
    # predictive draws share the weight draws, so points correlate; noise-only part: compare
    # the residual after removing the noiseless prediction of the SAME selection is not
    # available here, so only sanity-check moments
    assert abs(z.mean()) < 1e-10 and abs((z ** 2).mean() - 1) < 1e-3
    # same seed -> same draws; other seed -> different
    a, _ = rndm_m_random_calculator(g["preds"], g["samples"], g["Vt_hat"], seed=5)
    b, _ = rndm_m_random_calculator(g["preds"], g["samples"], g["Vt_hat"], seed=5)
    c2, _ = rndm_m_random_calculator(g["preds"], g["samples"], g["Vt_hat"], seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c2)


def test_noise_is_standard_normal():
    """Zero weights isolate the generator: rndm_m = 1/Km * sum(preds) + z * sigma."""
    ctx = gpu_ctx()
    M, Km, k, S = 96, 4, 2, 10000
    preds = np.zeros((M, Km))
    theta = np.column_stack([np.zeros((S, k)), np.full(S, 2.0)])
    Vt = np.zeros((k, Km))
    rndm_m, _, _ = ctx.predict(preds, theta, Vt, seed=11, q=())
    z = rndm_m / 2.0
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 5 * np.sqrt(2 / z.size)
    c = np.corrcoef(z.T)
    off = c[~np.eye(M, dtype=bool)]
    assert np.abs(off).max() < 6 / np.sqrt(S)
    from scipy import stats
    assert stats.kstest(z[:, :8].ravel(), "norm").pvalue > 1e-4


def test_too_few_samples_raises_like_the_reference():
    g = load_golden("predict_synth48")
    with pytest.raises(ValueError):
        rndm_m_random_calculator(g["preds"], g["samples"][:9999], g["Vt_hat"])


def make_bmc():
    rng = np.random.Generator(np.random.PCG64(17))
    truth = rng.standard_normal(200) * 3 + 10
    cols = {"N": np.arange(200), "Z": np.arange(200) % 17, "truth": truth}
    for j in range(6):
        cols[f"m{j}"] = truth + rng.normal(0.3 * j, 1.0, 200)
    df = pd.DataFrame(cols)
    models = [f"m{j}" for j in range(6)]
    return BayesianModelCombination(models, {"BE": df}, "truth"), df, models


def test_bmc_predict_predict2_evaluate():  # reference tests/test_bmc.py:122-220
    bmc, df, models = make_bmc()
    bmc.orthogonalize("BE", df.iloc[:150], 4)
    bmc.train({"iterations": 12000, "seeds": [3]})
    rndm_m, lower_df, median_df, upper_df = bmc.predict(df.iloc[150:][["N", "Z"] + models])
    assert isinstance(rndm_m, np.ndarray) and rndm_m.shape == (10000, 50)
    for f, col in ((lower_df, "Predicted_Lower"), (median_df, "Predicted_Median"),
                   (upper_df, "Predicted_Upper")):
        assert isinstance(f, pd.DataFrame) and col in f.columns and "N" in f.columns
    assert (lower_df["Predicted_Lower"] < upper_df["Predicted_Upper"]).all()
    r2, lo2, med2, up2 = bmc.predict2("BE")
    assert r2.shape == (10000, 200) and "truth" not in med2.columns
    # the combination predicts the truth better than the crude models (noise sd 1 + bias)
    err = np.abs(med2["Predicted_Median"].to_numpy() - df["truth"].to_numpy())
    assert err.mean() < 0.6
    cov = bmc.evaluate()
    assert isinstance(cov, list) and len(cov) == 21 and cov[0] == 0.0
    assert all(b >= a - 1e-9 for a, b in zip(cov, cov[1:])) and cov[-1] > 95
    cov_f = bmc.evaluate(domain_filter={"Z": (0, 8)})
    assert len(cov_f) == 21
    with pytest.raises(KeyError):
        bmc.predict2("nope")
    with pytest.raises(ValueError):
        bmc.predict("not a frame")
