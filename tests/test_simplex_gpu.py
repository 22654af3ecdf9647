"""Simplex-constrained sampler on the GPU (-m gpu): reference inference_utils.py:59-144.

Replay tier: the reference's own runs (golden fixtures; the proposals, the conditionally
consumed uniforms and the gamma stream were recorded from outside) are reproduced through
the HIP kernel.  Tolerance: 1e-9 absolute on the chain (observed ~1e-15), equality on the
acceptance count and on the number of uniforms consumed.
"""
import numpy as np
import pytest

from conftest import load_golden
from gpu_common import gpu_ctx
from oracle import bmc_oracle as O
from pybmc_amd import gibbs_sampler_simplex

pytestmark = pytest.mark.gpu


def replay_inputs(g):
    burn, T = int(g["burn"]), g["samples"].shape[0]
    step = g["S_hat"] * float(g["stepsize"])
    props = g["proposals"]
    means = []
    it = iter(props)
    real = O.mvn_draw_svd
    O.mvn_draw_svd = lambda mean, cov, z: next(it)
    try:
        out, acc, used = O.simplex_replay(
            g["y"], g["X"], g["Vt_hat"], g["S_hat"], T, [float(g["nu0"]), float(g["s20"])], burn,
            float(g["stepsize"]), np.zeros((burn + T, g["X"].shape[1])), g["uniforms"], g["G"],
            means_out=means)
    finally:
        O.mvn_draw_svd = real
    assert np.array_equal(out, g["samples"])
    xi = (props - np.array(means)) / step[None, :]
    return burn, T, xi, acc, used


@pytest.mark.parametrize("name", ["simplex_tiny3x2", "simplex_synth150x4"])
def test_replay_matches_the_reference(name):
    g = load_golden(name)
    ctx = gpu_ctx()
    burn, T, xi, acc, used = replay_inputs(g)
    ctx.set_problem(g["y"], g["X"])
    out, accepted, used_dev, st = ctx.simplex_run(
        g["Vt_hat"], g["S_hat"], T, float(g["nu0"]), float(g["s20"]), burn, float(g["stepsize"]),
        xi=xi, unif=g["uniforms"], g=g["G"], return_stats=True)
    assert np.abs(out - g["samples"]).max() < 1e-9
    assert accepted == acc and used_dev == used == len(g["uniforms"])
    # sizes like the reference's run in ONE wave (simplex_wave_kernel); the workgroup form gives
    # the same chain (kept rows are staged 64 at a time: T is not a multiple of 64 here)
    assert st["waves_per_group"] == 1 and st["groups_per_chain"] == 1, st
    ctx.set_tuning(groups_per_chain=1, waves_per_group=2)
    out2, accepted2, used2, st2 = ctx.simplex_run(
        g["Vt_hat"], g["S_hat"], T, float(g["nu0"]), float(g["s20"]), burn, float(g["stepsize"]),
        xi=xi, unif=g["uniforms"], g=g["G"], return_stats=True)
    ctx.set_tuning()
    assert st2["waves_per_group"] == 2
    assert accepted2 == accepted and used2 == used_dev
    assert np.abs(out2 - out).max() < 1e-12


def test_replay_is_geometry_independent():
    g = load_golden("simplex_synth150x4")
    ctx = gpu_ctx()
    burn, T, xi, acc, used = replay_inputs(g)
    ctx.set_problem(g["y"], g["X"])
    for G, W, res, ppw, agent in [(1, 3, 1, 1, 0), (3, 1, 1, 1, 1), (2, 2, 2, 0, 0), (1, 4, 3, 0, 0),
                                  (1, 1, 0, 0, 0)]:
        ctx.set_tuning(G, W, res, ppw, agent)
        out, accepted = ctx.simplex_run(g["Vt_hat"], g["S_hat"], T, float(g["nu0"]), float(g["s20"]),
                                        burn, float(g["stepsize"]), xi=xi, unif=g["uniforms"], g=g["G"])
        assert np.abs(out - g["samples"]).max() < 1e-9 and accepted == acc
    ctx.set_tuning(0, 0)


@pytest.mark.parametrize("n,k,km,nw", [(2500, 3, 4, 4), (1024, 8, 9, 2), (4000, 4, 6, 4)])
def test_chain_in_two_or_four_waves_matches_the_workgroup_form(n, k, km, nw):
    """A few thousand rows: the simplex sampler runs in 2 or 4 register-resident waves of one
    workgroup (simplex_wave_kernel, every wave running the whole step).  Same variates, same
    decisions, same chain as the workgroup form (simplex_loop_kernel)."""
    ctx = gpu_ctx()
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, km))
    truth = A @ np.full(km, 1.0 / km) + 0.05 * rng.standard_normal(n)
    Ac = A - A.mean(1, keepdims=True)
    U, S, Vt = np.linalg.svd(Ac, full_matrices=False)
    X = U[:, :k]
    S_hat = S[:k]
    Vt_hat = Vt[:k] / S_hat[:, None]
    y = truth - A.mean(1)
    ctx.set_problem(y, X)
    burn, T = 300, 1000
    out, acc, used, st = ctx.simplex_run(Vt_hat, S_hat, T, 1.0, 0.02, burn, 0.001, seed=5, return_stats=True)
    assert st["waves_per_group"] == nw and st["groups_per_chain"] == 1, st
    ctx.set_tuning(waves_per_group=8)
    ref, acc2, used2, st2 = ctx.simplex_run(Vt_hat, S_hat, T, 1.0, 0.02, burn, 0.001, seed=5, return_stats=True)
    ctx.set_tuning()
    assert st2["waves_per_group"] == 8
    assert acc == acc2 and used == used2 and 0 < acc
    assert np.abs(out - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    w = out[:, :k] @ Vt_hat + 1.0 / km
    assert (w >= 0).all()


def test_reference_surface():  # reference tests/test_inference_utils.py:21-32, 47-76
    y = np.array([1.0, 2.0, 3.0])
    X = np.array([[1, 0], [0, 1], [1, 1]])
    Vt_hat = np.array([[0.5, 0.5], [0.5, -0.5]])
    S_hat = np.array([1.0, 0.5])
    samples = gibbs_sampler_simplex(y, X, Vt_hat, S_hat, 10, [1.0, 1.0], burn=100, stepsize=0.01)
    assert samples.shape == (10, 3) and not np.any(np.isnan(samples))
    with pytest.raises(ValueError):
        gibbs_sampler_simplex(y, X, Vt_hat, S_hat, 10, [1.0, 1.0], burn=-1)
    with pytest.raises(ValueError):
        gibbs_sampler_simplex(y, X, Vt_hat, S_hat, 10, [1.0, 1.0], stepsize=-0.01)
    s = gibbs_sampler_simplex(y, X, Vt_hat, S_hat, 100, [1.0, 1.0], burn=10, stepsize=0.01)
    assert 0 < len(s) <= 100 and not np.any(np.isnan(s))
    # weights stay on the simplex
    w = s[:, :2] @ Vt_hat + 0.5
    assert (w >= 0).all()


def test_device_generator_distribution():
    """Free-running device chain vs the oracle driven by numpy streams: same acceptance
    rate and posterior means within Monte-Carlo error."""
    g = load_golden("simplex_synth150x4")
    ctx = gpu_ctx()
    ctx.set_problem(g["y"], g["X"])
    burn, T = 2000, 40000
    k = g["X"].shape[1]
    out, accepted = ctx.simplex_run(g["Vt_hat"], g["S_hat"], T, 1.0, 0.02, burn, 0.002, seed=7)
    rs = np.random.RandomState(1)
    Z = rs.standard_normal((burn + T, k))
    U = rs.uniform(size=burn + T)
    G = np.random.Generator(np.random.PCG64(2)).standard_gamma((1.0 + len(g["y"])) / 2, size=burn + T)
    real = O.mvn_draw_svd
    step = g["S_hat"] * 0.002
    O.mvn_draw_svd = lambda mean, cov, z: mean + step * z      # the diagonal map, any sign pattern
    try:
        ref, acc_ref, _ = O.simplex_replay(g["y"], g["X"], g["Vt_hat"], g["S_hat"], T, [1.0, 0.02],
                                           burn, 0.002, Z, U, G)
    finally:
        O.mvn_draw_svd = real
    assert abs(accepted / T - acc_ref / T) < 0.02
    # strongly autocorrelated random walk: compare with a generous batch-means error
    def bm(x):
        m = x[: len(x) // 40 * 40].reshape(40, -1, x.shape[1]).mean(1)
        return m.std(0, ddof=1) / np.sqrt(40)
    se = np.sqrt(bm(out) ** 2 + bm(ref) ** 2)
    assert np.all(np.abs(out.mean(0) - ref.mean(0)) < 6 * se + 1e-12)
    w = out[:, :k] @ g["Vt_hat"] + 1.0 / g["Vt_hat"].shape[1]
    assert (w >= 0).all()


def test_bmc_train_simplex():  # reference tests/test_bmc.py:107-120
    import pandas as pd
    from pybmc_amd import BayesianModelCombination
    data = {"property": pd.DataFrame({"model1": [1, 2], "model2": [3, 4], "truth": [5, 6]})}
    bmc = BayesianModelCombination(["model1", "model2"], data, "truth")
    bmc.orthogonalize("property", pd.DataFrame({"model1": [1, 2], "model2": [3, 4], "truth": [5, 6]}), 1)
    bmc.train({"iterations": 100, "sampler": "simplex", "burn": 10, "stepsize": 0.01,
               "b_mean_prior": np.zeros(1), "b_mean_cov": np.eye(1), "nu0_chosen": 1.0,
               "sigma20_chosen": 0.02})
    assert bmc.samples is not None and bmc.samples.shape[0] == 100
