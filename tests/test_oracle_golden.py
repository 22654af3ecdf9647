"""Pins the CPU oracle to outputs of the unmodified reference (bit-for-bit).

The fixtures come from tests/golden/make_golden.py, which imports the reference
with both of its random streams pinned from the outside.  These tests are the
"oracle pinned" evidence: every oracle function is compared with what the
reference itself produced, tolerance 0.
"""
import numpy as np
import pytest

from oracle import bmc_oracle as O
from pybmc_amd.synthetic import synth_problem, sha256
from conftest import load_golden

GIBBS_CASES = ["gibbs_tiny3x2", "gibbs_dense64x8", "gibbs_ortho629x3",
               "gibbs_ragged1237x5", "gibbs_c2_10000x32"]


def case_inputs(g):
    if "X" in g:
        X, y = g["X"], g["y"]
    else:
        n, km, kept, seed = (int(v) for v in g["synth"])
        p = synth_problem(n, km, kept, seed)
        X, y = p["X"], p["y"]
    assert sha256(np.asarray(X, float, order="F")) == str(g["X_sha"])
    assert sha256(np.asarray(y, float)) == str(g["y_sha"])
    prior = (g["b0"], g["C0"], float(g["nu0"]), float(g["s20"]))
    return y, X, prior


@pytest.mark.parametrize("name", GIBBS_CASES)
def test_gibbs_replay_is_the_reference(name):
    g = load_golden(name)
    y, X, prior = case_inputs(g)
    T = int(g["T"])
    if name == "gibbs_c2_10000x32":
        T = 120  # keep the CPU suite short; the chain prefix is what is compared
    out = O.gibbs_replay(y, X, T, prior, g["Z"], g["G"])
    assert np.array_equal(out, g["samples"][:T])


@pytest.mark.parametrize("name", GIBBS_CASES[:4])
def test_streams_regenerate_from_seeds(name):
    g = load_golden(name)
    y, X, prior = case_inputs(g)
    st = O.chain_setup(y, X, prior)
    Z, G = O.reference_streams(int(g["seed_z"]), int(g["seed_g"]), int(g["T"]),
                               X.shape[1], O.gamma_shape(st))
    assert np.array_equal(Z, g["Z"]) and np.array_equal(G, g["G"])


def test_gibbs_port_matches_when_pinned():
    """The timed CPU baseline makes the same numpy calls as the reference:
    with the two streams pinned the same way it reproduces the golden chain."""
    g = load_golden("gibbs_dense64x8")
    y, X, prior = case_inputs(g)
    gen = np.random.Generator(np.random.PCG64(int(g["seed_g"])))
    real = np.random.default_rng
    np.random.default_rng = lambda *a, **k: gen if not (a or k) else real(*a, **k)
    try:
        np.random.seed(int(g["seed_z"]))
        out = O.gibbs_port(y, X, 200, prior)
    finally:
        np.random.default_rng = real
    assert np.array_equal(out, g["samples"][:200])


def test_usvt_and_centring_testbmc_frame():
    g = load_golden("ortho_testbmc")
    F = np.array([[10, 15, 12], [20, 25, 30], [30, 35, 32], [40, 45, 43]], float)
    truth = np.array([11, 21, 31, 41], float)
    mu, yc, U_hat, S_hat, Vt_hat, Vt_n = O.centre_and_svd(F, truth, 2)
    assert np.array_equal(mu, g["predictions_mean_train"])
    assert np.array_equal(yc, g["centered_experiment_train"])
    for a, b in ((U_hat, "U_hat"), (S_hat, "S_hat"), (Vt_hat, "Vt_hat"),
                 (Vt_n, "Vt_hat_normalized")):
        assert np.array_equal(a, g[b])
    assert U_hat.flags.f_contiguous


def test_centring_synth_frame():
    g = load_golden("ortho_synth200x6")
    fr = g["frame"]
    F, truth = fr[:150, 3:], fr[:150, 2]
    mu, yc, U_hat, S_hat, Vt_hat, Vt_n = O.centre_and_svd(F, truth, 4)
    assert np.array_equal(U_hat, g["U_hat"]) and np.array_equal(Vt_hat, g["Vt_hat"])
    assert np.array_equal(yc, g["centered_experiment_train"])
    # thin SVD gives the same leading columns up to rounding (what the product uses)
    _, _, U2, S2, V2, _ = O.centre_and_svd(F, truth, 4, full_matrices=False)
    np.testing.assert_allclose(U2, U_hat, rtol=0, atol=1e-13)
    np.testing.assert_allclose(S2, S_hat, rtol=1e-13)


def test_predictive_and_coverage():
    g = load_golden("predict_synth48")
    rng = np.random.Generator(np.random.PCG64(int(g["seed_g"])))
    rndm_m, (lo, med, up) = O.predictive_replay(g["preds"], g["samples"], g["Vt_hat"], rng)
    assert sha256(rndm_m) == str(g["rndm_m_sha"])
    assert np.array_equal(lo, g["lower"]) and np.array_equal(med, g["median"])
    assert np.array_equal(up, g["upper"])
    cov = O.coverage_oracle(np.arange(0, 101, 5), rndm_m, g["truth"])
    assert np.array_equal(np.array(cov), g["coverage"])


@pytest.mark.parametrize("name", ["simplex_tiny3x2", "simplex_synth150x4"])
def test_simplex_replay(name):
    g = load_golden(name)
    S_hat, step = g["S_hat"], float(g["stepsize"])
    burn = int(g["burn"])
    T = g["samples"].shape[0]
    # recover the proposal normals from the recorded proposals: the step
    # covariance is diagonal, so its svd map is a signed permutation; instead of
    # inverting it the oracle is fed the recorded proposals directly.
    out = replay_simplex_from_proposals(g, burn, T)
    assert np.array_equal(out, g["samples"])


def replay_simplex_from_proposals(g, burn, T):
    """simplex_replay with the proposal draw short-circuited to the recorded one
    (the legacy stream interleaves normals and conditionally-consumed uniforms,
    so the recorded values, not seeds, are the fixture)."""
    real = O.mvn_draw_svd
    props = iter(g["proposals"])
    O.mvn_draw_svd = lambda mean, cov, z: next(props)
    try:
        out, acc, used = O.simplex_replay(
            g["y"], g["X"], g["Vt_hat"], g["S_hat"], T, [float(g["nu0"]), float(g["s20"])],
            burn, float(g["stepsize"]), np.zeros((burn + T, g["X"].shape[1])),
            g["uniforms"], g["G"])
    finally:
        O.mvn_draw_svd = real
    assert used == len(g["uniforms"])
    return out
