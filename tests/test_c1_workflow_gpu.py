"""BASELINE.json configs[0] (-m gpu): the docs/usage.md workflow on a stand-in for the
nuclear-mass table (selected_data.h5 is not available: SURVEY.md 8d -- these are NOT nuclear
data): 629 rows, domain columns N and Z, a truth column and 4 models, random 0.6/0.2/0.2 split,
3 components kept, 1 chain x 2000 iterations.  The reference-CPU side of the comparison is the
oracle (numpy restatement, pinned to the reference) driven by numpy's own streams; the device
replays that chain and must reproduce the posterior weight means to 1e-6 relative."""
import numpy as np
import pandas as pd
import pytest

from gpu_common import gpu_ctx
from oracle import bmc_oracle as O
from pybmc_amd import BayesianModelCombination, Dataset
from pybmc_amd.chains import posterior_summary

pytestmark = pytest.mark.gpu

MODELS = ["FRDM12", "HFB24", "D1M", "UNEDF1"]


def stand_in_csv(path):
    rng = np.random.Generator(np.random.PCG64(0))
    n = 629
    N = rng.integers(8, 160, n)
    Z = rng.integers(8, 100, n)
    truth = 8.0 * (N + Z) - 0.01 * (N - Z) ** 2 + rng.standard_normal(n)
    rows = []
    for j, m in enumerate(MODELS + ["AME2020"]):
        be = truth if m == "AME2020" else truth + rng.normal(0.4 * j - 0.5, 1.0, n) + rng.normal(0, 0.5, n)
        rows.append(pd.DataFrame({"N": N, "Z": Z, "BE": be, "model": m}))
    pd.concat(rows).to_csv(path, index=False)


def test_usage_workflow_matches_the_cpu_reference_path(tmp_path):
    csv = tmp_path / "stand_in.csv"
    stand_in_csv(csv)
    ds = Dataset(str(csv))
    data = ds.load_data(models=MODELS + ["AME2020"], keys=["BE"], domain_keys=["N", "Z"])
    assert len(data["BE"]) > 500
    train_df, val_df, test_df = ds.split_data(data, "BE", splitting_algorithm="random",
                                              train_size=0.6, val_size=0.2, test_size=0.2)
    bmc = BayesianModelCombination(MODELS, data, truth_column_name="AME2020")
    bmc.orthogonalize("BE", train_df, components_kept=3, method="svd")
    T = 2000
    prior = (np.zeros(3), np.diag(bmc.S_hat ** 2), 1.0, 0.02)        # train() defaults, bmc.py:168-171

    # CPU reference path: the reference's sampler restated, fed by numpy's own streams
    y, X = bmc.centered_experiment_train, bmc.U_hat
    st = O.chain_setup(y, X, prior)
    Z, G = O.reference_streams(7, 8, T, 3, O.gamma_shape(st))
    ref, trace = O.gibbs_replay(y, X, T, prior, Z, G, return_sigma2=True)

    # device: the same chain replayed through the HIP loop
    ctx = gpu_ctx()
    ctx.set_problem(y, X)
    ctx.set_prior(*prior)
    W, lam, _ = ctx.basis()
    xi = O.innovations_in_basis(st, y, X, ref, W, lam, trace)
    out, stats = ctx.gibbs_run(1, T, xi=xi[None], g=G[None])
    a, b = posterior_summary(out[0], bmc.Vt_hat), posterior_summary(ref, bmc.Vt_hat)
    for key in b:
        err = np.abs(a[key] - b[key]).max() / np.abs(b[key]).max()
        assert err < 1e-6, (key, err)
    assert abs(a["weights_mean"].sum() - 1) < 1e-9

    # and the public API end to end (its own Philox stream): statistically the same posterior
    bmc.train({"iterations": T, "sampler": "gibbs_sampling"})
    assert bmc.samples.shape == (T, 4)
    w_api = posterior_summary(bmc.samples, bmc.Vt_hat)["weights_mean"]
    assert np.abs(w_api - b["weights_mean"]).max() < 0.05
    # predict*/evaluate need >= 10000 samples (sampling_utils.py:57), like the reference
    with pytest.raises(ValueError):
        bmc.predict2("BE")
    bmc.train({"iterations": 12000})
    rndm_m, lower_df, median_df, upper_df = bmc.predict2("BE")
    assert rndm_m.shape[0] == 10000 and list(median_df.columns) == ["N", "Z", "Predicted_Median"]
    cov = bmc.evaluate()
    assert len(cov) == 21 and cov[19] > 80
