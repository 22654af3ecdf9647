"""orthogonalize on the device (-m gpu): reference bmc.py:106-122 + inference_utils.py:147-168
through bmc_orthogonalize (centre -> f64-MFMA Gram -> K x K eigen -> U_hat = Fc V S^-1).

LAPACK's singular-vector signs are arbitrary, so vectors are compared up to one sign per
component; everything the sampler and the predictive use (S_hat, |U_hat|, weights) must
agree with the reference's values (golden fixture) to 1e-10."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from gpu_common import gpu_ctx
from pybmc_amd import BayesianModelCombination

pytestmark = pytest.mark.gpu


def align(U, V_rows, U_ref):
    sg = np.sign(np.sum(U * U_ref, axis=0))
    return U * sg, V_rows * sg[:, None]


def test_matches_reference_golden():
    g = load_golden("ortho_synth200x6")
    fr = g["frame"]
    F, truth = fr[:150, 3:], fr[:150, 2]
    ctx = gpu_ctx()
    mu, yc, U, S, Vt = ctx.orthogonalize(F, truth, 4)
    assert np.abs(mu - g["predictions_mean_train"]).max() < 1e-13
    assert np.abs(yc - g["centered_experiment_train"]).max() < 1e-13
    U, Vt = align(U, Vt, g["U_hat"])
    assert np.abs(S - g["S_hat"]).max() < 1e-11 * g["S_hat"][0]
    assert np.abs(U - g["U_hat"]).max() < 1e-10
    assert np.abs(Vt - g["Vt_hat_normalized"]).max() < 1e-10
    assert np.abs(Vt / S[:, None] - g["Vt_hat"]).max() < 1e-10
    # largest entry of each right singular vector is positive (the library's convention)
    _, _, _, _, Vraw = ctx.orthogonalize(F, truth, 4)
    assert all(v[np.argmax(np.abs(v))] > 0 for v in Vraw)
    # the context now holds (y_c, U_hat): its Gram is [I  U'y; y'U  y'y]
    G = ctx.gram()
    assert np.abs(G[:4, :4] - np.eye(4)).max() < 1e-12


def test_null_space_is_refused():
    g = load_golden("ortho_synth200x6")
    fr = g["frame"]
    ctx = gpu_ctx()
    with pytest.raises(np.linalg.LinAlgError):
        ctx.orthogonalize(fr[:150, 3:], fr[:150, 2], 6)      # rank <= n_models - 1
    with pytest.raises(ValueError):
        ctx.orthogonalize(fr[:150, 3:], fr[:150, 2], 7)


@pytest.mark.parametrize("n,km,k", [(1000, 5, 3), (50000, 33, 32), (200000, 65, 64)])
def test_large_sizes(n, km, k):
    rng = np.random.Generator(np.random.PCG64(n))
    F = rng.standard_normal((n, km)) + rng.standard_normal(n)[:, None] * 3
    truth = F.mean(1) + rng.standard_normal(n)
    ctx = gpu_ctx()
    mu, yc, U, S, Vt = ctx.orthogonalize(F, truth, k)
    Fc = F - F.mean(1)[:, None]
    S_ref = np.linalg.svd(Fc, compute_uv=False)[:k]
    assert np.abs(S - S_ref).max() < 1e-10 * S_ref[0]
    assert np.abs(U.T @ U - np.eye(k)).max() < 1e-9
    assert np.abs(Vt @ Vt.T - np.eye(k)).max() < 1e-10
    # U S Vt reproduces the projection of Fc on the kept right singular vectors
    assert np.abs(U * S - Fc @ Vt.T).max() < 1e-9 * S_ref[0]
    assert np.abs(Vt.sum(1)).max() < 1e-9          # kept vectors are orthogonal to 1


def test_bmc_device_route_end_to_end():
    rng = np.random.Generator(np.random.PCG64(5))
    n = 30000
    truth = rng.standard_normal(n) * 3 + 10
    cols = {"N": np.arange(n), "truth": truth}
    for j in range(6):
        cols[f"m{j}"] = truth + rng.normal(0.3 * j, 1.0, n)
    df = pd.DataFrame(cols)
    models = [f"m{j}" for j in range(6)]
    a = BayesianModelCombination(models, {"BE": df}, "truth")
    b = BayesianModelCombination(models, {"BE": df}, "truth")
    a.orthogonalize("BE", df, 4, method="device")
    b.orthogonalize("BE", df, 4, method="svd")
    assert a._device_problem is not None and a.U_hat.flags.f_contiguous
    sg = np.sign(np.sum(a.U_hat * b.U_hat, axis=0))
    assert np.abs(a.U_hat * sg - b.U_hat).max() < 1e-9
    a.train({"iterations": 12000, "seeds": [1]})
    b.train({"iterations": 12000, "seeds": [1]})
    wa = a.samples[2000:, :4].mean(0) @ a.Vt_hat + 1 / 6
    wb = b.samples[2000:, :4].mean(0) @ b.Vt_hat + 1 / 6
    assert abs(wa.sum() - 1) < 1e-9
    # same seed, sign-flipped basis -> same chain up to the sign map; the weights agree closely
    assert np.abs(wa - wb).max() < 5e-3
    cov = a.evaluate()
    assert len(cov) == 21 and cov[-1] > 95
