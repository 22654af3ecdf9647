"""Opt-in rss_mode 1 (-m gpu): rss from sufficient statistics instead of a pass over the data.

Same parity bar as the data-pass loop: the reference's own chains (golden fixtures) replayed
through the kernel with its innovations in the build's basis, 1e-9 relative on every row; and
the data-pass loop itself as a second witness on Philox-driven chains."""
import numpy as np
import pytest

from gpu_common import golden_case, gpu_ctx, replay_inputs
from pybmc_amd import gibbs_sampler
from pybmc_amd._lib import BmcError

pytestmark = pytest.mark.gpu

FIXTURES = ["gibbs_tiny3x2", "gibbs_dense64x8", "gibbs_ortho629x3", "gibbs_ragged1237x5",
            "gibbs_c2_10000x32"]


@pytest.mark.parametrize("name", FIXTURES)
def test_replay_matches_the_reference(name):
    ctx = gpu_ctx()
    g, y, X, prior = golden_case(name)
    ctx.set_problem(y, X)
    ctx.set_prior(*prior)
    T = int(g["T"])
    st, xi, ref = replay_inputs(ctx, g, y, X, prior, T)
    G = g["G"][:T]
    ctx.set_tuning(rss_mode=1)
    try:
        out, stats = ctx.gibbs_run(3, T, xi=np.repeat(xi[None], 3, 0), g=np.repeat(G[None], 3, 0))
    finally:
        ctx.set_tuning()
    assert stats["residency"] == 4 and stats["passes"] == 0 and stats["launches"] == 1
    for c in range(3):
        assert np.abs(out[c] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n,k,snr", [(10000, 32, 10.0), (5000, 64, 1e4), (700, 1, 3.0), (90, 17, 1e6)])
def test_same_chain_as_the_data_pass(n, k, snr):
    """Philox-driven chains, rss by data pass vs from sufficient statistics, including fits so
    tight that rss is 1e-12 of |y|^2 (where y'y - 2u'X'y + u'X'Xu would cancel to nothing)."""
    ctx = gpu_ctx()
    rng = np.random.default_rng(n + k)
    X = rng.standard_normal((n, k)) / np.sqrt(n)
    beta = rng.standard_normal(k)
    sig = np.linalg.norm(X @ beta) / np.sqrt(n) / snr
    y = X @ beta + sig * rng.standard_normal(n)
    ctx.set_problem(y, X)
    ctx.set_prior(np.zeros(k), np.eye(k) * 10.0, 1.0, 0.02 * sig ** 2)
    seeds = np.arange(5) + 3
    data, _ = ctx.gibbs_run(5, 400, seeds=seeds)
    ctx.set_tuning(rss_mode=1)
    try:
        gram, st = ctx.gibbs_run(5, 400, seeds=seeds)
    finally:
        ctx.set_tuning()
    assert st["residency"] == 4
    scale = np.abs(data).max(axis=(0, 1))
    assert (np.abs(gram - data).max(axis=(0, 1)) < 1e-8 * scale).all()
    if sig ** 2 > 1e-5:   # otherwise the 1e-6 floor on sigma2 (:52) binds, in both modes
        assert abs(gram[..., -1].mean() / sig - 1) < 0.1


def test_many_chains_in_one_launch_and_the_python_surface():
    rng = np.random.default_rng(0)
    n, k = 2000, 8
    X = rng.standard_normal((n, k)) / np.sqrt(n)
    y = X @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)
    prior = (np.zeros(k), np.eye(k) * 10.0, 1.0, 0.02)
    out, st = gibbs_sampler(y, X, 300, prior, n_chains=700, seeds=np.arange(700), rss="gram",
                            return_stats=True)
    assert out.shape == (700, 300, k + 1) and st["launches"] == 1 and np.isfinite(out).all()
    solo = gibbs_sampler(y, X, 300, prior, seeds=[123], rss="gram")
    assert np.array_equal(solo, out[123])
    # the next call without the option is the data-pass loop again
    _, st2 = gibbs_sampler(y, X, 50, prior, seeds=[1], return_stats=True)
    assert st2["residency"] in (1, 2, 3)
    with pytest.raises(ValueError):
        gibbs_sampler(y, X, 10, prior, rss="fast")


def test_more_than_64_columns_is_refused():
    ctx = gpu_ctx()
    rng = np.random.default_rng(1)
    X = rng.standard_normal((500, 65))
    y = rng.standard_normal(500)
    ctx.set_problem(y, X)
    ctx.set_prior(np.zeros(65), np.eye(65), 1.0, 1.0)
    ctx.set_tuning(rss_mode=1)
    try:
        with pytest.raises((BmcError, ValueError)):
            ctx.gibbs_run(1, 10, seeds=[1])
    finally:
        ctx.set_tuning()
