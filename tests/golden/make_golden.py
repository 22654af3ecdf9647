#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Runs only in the build container (it imports /root/reference, which never
travels to the GPU box).  Nothing is copied from the reference: the files
written here are arrays it *produced* (inputs + outputs), stored as .npz with
``allow_pickle=False``-loadable content only.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

How the reference is pinned without editing it:
the beta draw uses the legacy global RandomState -> ``np.random.seed(seed_z)``;
the sigma2 draw builds a fresh unseeded ``np.random.default_rng()`` every
iteration -> for the duration of the call that name is pointed at a function
returning ONE persistent ``Generator(PCG64(seed_g))``.
"""
import contextlib
import hashlib
import os
import sys

import numpy as np

REF = os.environ.get("PYBMC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import pandas as pd  # noqa: E402
from pybmc.inference_utils import (  # noqa: E402
    gibbs_sampler, gibbs_sampler_simplex, USVt_hat_extraction)
from pybmc.sampling_utils import coverage, rndm_m_random_calculator  # noqa: E402
from pybmc.bmc import BayesianModelCombination  # noqa: E402
from pybmc.data import Dataset  # noqa: E402


@contextlib.contextmanager
def pinned(seed_z, seed_g):
    gen = np.random.Generator(np.random.PCG64(seed_g))
    real = np.random.default_rng

    def fake(*a, **k):
        if a or k:
            return real(*a, **k)
        return gen

    np.random.default_rng = fake
    np.random.seed(seed_z)
    try:
        yield gen
    finally:
        np.random.default_rng = real


@contextlib.contextmanager
def recording_legacy():
    """Wrap (not edit) the legacy draws so the values consumed can be stored."""
    log = {"mvn": [], "unif": []}
    mvn, unif = np.random.multivariate_normal, np.random.uniform

    def mvn_w(mean, cov, *a, **k):
        out = mvn(mean, cov, *a, **k)
        log["mvn"].append((np.array(mean, dtype=float), np.array(out)))
        return out

    def unif_w(*a, **k):
        out = unif(*a, **k)
        log["unif"].append(out)
        return out

    np.random.multivariate_normal, np.random.uniform = mvn_w, unif_w
    try:
        yield log
    finally:
        np.random.multivariate_normal, np.random.uniform = mvn, unif


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth_problem(n, k_models, kept, seed, noise=0.1):
    """SURVEY 8d synthetic recipe: gaussian model matrix, rows centred, thin SVD,
    the first `kept` left singular vectors are X (column-major), prior = the
    train() defaults (bmc.py:168-171)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    F = rng.standard_normal((n, k_models))
    Fc = F - F.mean(axis=1)[:, None]
    U, S, Vt = np.linalg.svd(Fc, full_matrices=False)
    X = np.asfortranarray(U[:, :kept])
    S_hat = S[:kept]
    beta_true = rng.standard_normal(kept)
    y = X @ beta_true + noise * rng.standard_normal(n)
    prior = (np.zeros(kept), np.diag(S_hat ** 2), 1.0, 0.02)
    return F, X, y, S_hat, Vt[:kept], prior


def gibbs_case(name, y, X, prior, T, seed_z, seed_g, store_inputs=True, extra=None):
    with pinned(seed_z, seed_g):
        samples = gibbs_sampler(y, X, T, prior)
    with pinned(seed_z, seed_g):
        again = gibbs_sampler(y, X, T, prior)
    assert np.array_equal(samples, again), "pinned reference is not reproducible"
    K = X.shape[1]
    shape = (prior[2] + len(y)) / 2.0
    Z = np.random.RandomState(seed_z).standard_normal((T, K))
    G = np.random.Generator(np.random.PCG64(seed_g)).standard_gamma(shape, size=T)
    d = dict(samples=samples, Z=Z, G=G, T=np.int64(T), seed_z=np.int64(seed_z),
             seed_g=np.int64(seed_g), b0=np.asarray(prior[0], float),
             C0=np.asarray(prior[1], float), nu0=np.float64(prior[2]),
             s20=np.float64(prior[3]))
    if store_inputs:
        d["X"] = np.asarray(X)
        d["y"] = np.asarray(y, float)
    d["X_sha"] = np.array(sha(np.asarray(X, float, order="F")))
    d["y_sha"] = np.array(sha(np.asarray(y, float)))
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, samples.shape, "mean sigma", samples[:, -1].mean())


def standin_csv(path):
    """A small table shaped like the nuclear-mass file the reference's docs load
    (selected_data.h5 is absent from the reference checkout): long format, one row per
    (model, N, Z) with two properties.  Models cover slightly different (N, Z) sets so the
    inner join of load_data (pybmc/data.py:121-125) has something to drop."""
    rng = np.random.Generator(np.random.PCG64(2024))
    rows = []
    grid = [(n, z) for z in range(20, 32) for n in range(z, z + 12)]
    for mi, model in enumerate(["truth", "FRDM", "HFB24", "UNEDF1", "SKM"]):
        for (n, z) in grid:
            if (n * 7 + z * 3 + mi) % 11 == 0 and model != "truth":
                continue                                   # this model lacks that nucleus
            be = 8.5 * (n + z) - 0.1 * (n - z) ** 2
            rad = 1.2 * (n + z) ** (1.0 / 3.0)
            rows.append((model, n, z, round(be + (0.0 if mi == 0 else rng.normal(0.2 * mi, 0.8)), 6),
                         round(rad + (0.0 if mi == 0 else rng.normal(0.0, 0.02)), 6)))
    order = rng.permutation(len(rows))                     # rows in no particular order
    df = pd.DataFrame([rows[i] for i in order], columns=["model", "N", "Z", "BE", "Rad"])
    df.to_csv(path, index=False)
    return df


def dataset_case():
    """f4: the reference's Dataset (pybmc/data.py) on the committed stand-in CSV: loading and
    alignment (:30-129), the overview (:131-192), both splits (:247-330, random_state = 1),
    the distance split on its own (:194-245) and the subset filters (:332-374)."""
    path = os.path.join(HERE, "dataset_standin.csv")
    standin_csv(path)
    models = ["truth", "FRDM", "HFB24", "UNEDF1", "SKM"]
    ds = Dataset(path)
    data = ds.load_data(models, keys=["BE", "Rad"], domain_keys=["N", "Z"])
    out = {}
    for prop, df in data.items():
        out[f"load_{prop}_columns"] = np.array(list(df.columns))
        out[f"load_{prop}_values"] = df.to_numpy(float)
        out[f"load_{prop}_index"] = df.index.to_numpy()
    view = ds.view_data()
    out["view_properties"] = np.array(view["available_properties"])
    out["view_models"] = np.array(view["available_models"])
    out["view_model_BE"] = ds.view_data(model_name="FRDM")["BE"].to_numpy(float)
    out["view_series"] = ds.view_data("Rad", "SKM").to_numpy(float)
    tr, va, te = ds.split_data(data, "BE", splitting_algorithm="random",
                               train_size=0.6, val_size=0.2, test_size=0.2)
    out["random_train"], out["random_val"], out["random_test"] = (
        tr.index.to_numpy(), va.index.to_numpy(), te.index.to_numpy())
    out["random_train_values"] = tr.to_numpy(float)
    # the distance split measures against ALL columns of the frame (data.py:276), so it is
    # given a frame of the two domain columns
    dom = {"dom": data["BE"][["N", "Z"]]}
    stable = [(26, 24), (30, 28), (34, 30)]
    tr, va, te = ds.split_data(dom, "dom", splitting_algorithm="inside_to_outside",
                               stable_points=stable, distance1=2.0, distance2=4.5)
    out["dist_train"], out["dist_val"], out["dist_test"] = (
        tr.index.to_numpy(), va.index.to_numpy(), te.index.to_numpy())
    pts = [tuple(r) for r in data["BE"][["N", "Z"]].to_numpy()[:60]]
    a, b, c = ds.separate_points_distance_allSets(pts, stable, 1.5, 3.0)
    out["sep_a"], out["sep_b"], out["sep_c"] = np.array(a), np.array(b), np.array(c)
    subsets = {
        "range": dict(filters={"Z": (22, 27)}),
        "list": dict(filters={"N": [24, 25, 30, 41]}),
        "value": dict(filters={"Z": 25}),
        "callable": dict(filters={"N": lambda s: s % 2 == 0}),
        "multi": dict(filters={"multi": lambda r: r["N"] - r["Z"] >= 6, "Z": (21, 30)}),
        "models": dict(filters={"Z": (20, 24)}, models_to_include=["FRDM", "SKM", "nope"]),
    }
    for name, kw in subsets.items():
        sub = ds.get_subset("BE", **kw)
        out[f"subset_{name}_index"] = sub.index.to_numpy()
        out[f"subset_{name}_columns"] = np.array(list(sub.columns))
        out[f"subset_{name}_values"] = sub.to_numpy(float)
    np.savez_compressed(os.path.join(HERE, "dataset_standin.npz"), **out)
    print("dataset_standin", {k: v.shape for k, v in out.items() if k.startswith("load")})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        dataset_case()
        return
    dataset_case()
    # ---- G1: the reference's own tiny test (tests/test_inference_utils.py:6-14)
    y = np.array([1.0, 2.0, 3.0])
    X = np.array([[1, 0], [0, 1], [1, 1]])
    gibbs_case("gibbs_tiny3x2", y, X, (np.array([0.0, 0.0]), np.eye(2), 1.0, 1.0),
               T=400, seed_z=11, seed_g=12)

    # ---- G2: small synthetic, dense non-orthogonal X, dense prior covariance
    rng = np.random.Generator(np.random.PCG64(5))
    X = rng.standard_normal((64, 8))
    y = X @ rng.standard_normal(8) + 0.3 * rng.standard_normal(64)
    A = rng.standard_normal((8, 8))
    C0 = A @ A.T + 0.5 * np.eye(8)
    gibbs_case("gibbs_dense64x8", y, X, (0.1 * rng.standard_normal(8), C0, 3.0, 0.5),
               T=600, seed_z=21, seed_g=22)

    # ---- G3: notebook-sized orthogonalised problem (629 rows, 4 models, keep 3)
    F, X, y, S_hat, Vt, prior = synth_problem(629, 4, 3, seed=3)
    gibbs_case("gibbs_ortho629x3", y, X, prior, T=2000, seed_z=31, seed_g=32,
               extra=dict(S_hat=S_hat, Vt=Vt))

    # ---- G4: the headline size N=10000, K=32 (inputs regenerated from the seed;
    #          sha256 of X and y stored so a drifted generator is detected)
    F, X, y, S_hat, Vt, prior = synth_problem(10000, 33, 32, seed=0)
    gibbs_case("gibbs_c2_10000x32", y, X, prior, T=500, seed_z=41, seed_g=42,
               store_inputs=False,
               extra=dict(S_hat=S_hat, synth=np.array([10000, 33, 32, 0])))

    # ---- G5: ragged N (not a multiple of anything), K=5, float inputs with offset
    rng = np.random.Generator(np.random.PCG64(9))
    X = rng.standard_normal((1237, 5)) + 0.2
    y = X @ rng.standard_normal(5) + rng.standard_normal(1237)
    gibbs_case("gibbs_ragged1237x5", y, X, (np.ones(5), 4.0 * np.eye(5), 2.0, 1.5),
               T=300, seed_z=51, seed_g=52)

    # ---- U1: USVt_hat_extraction + orthogonalize on the test_bmc frame (:10-19)
    df = pd.DataFrame({
        "x": [1, 2, 3, 4, 5, 6], "y": [10, 11, 12, 13, 14, 15],
        "truth": [11, 21, 31, 41, 51, 61], "model1": [10, 20, 30, 40, 50, 60],
        "model2": [15, 25, 35, 45, 55, 65], "model3": [12, 30, 32, 43, 58, 67]})
    bmc = BayesianModelCombination(["model1", "model2", "model3", "truth"],
                                   {"target": df}, "truth")
    bmc.orthogonalize("target", df.iloc[:4], 2)
    with pinned(61, 62):
        bmc.train({"iterations": 300, "sampler": "gibbs_sampling", "burn": 0,
                   "stepsize": 0.001, "b_mean_prior": np.zeros(2),
                   "b_mean_cov": np.diag(bmc.S_hat ** 2), "nu0_chosen": 1.0,
                   "sigma20_chosen": 0.02})
    np.savez_compressed(
        os.path.join(HERE, "ortho_testbmc.npz"),
        U_hat=bmc.U_hat, S_hat=bmc.S_hat, Vt_hat=bmc.Vt_hat,
        Vt_hat_normalized=bmc.Vt_hat_normalized,
        centered_experiment_train=bmc.centered_experiment_train,
        predictions_mean_train=bmc._predictions_mean_train,
        samples=bmc.samples, seed_z=np.int64(61), seed_g=np.int64(62))
    print("ortho_testbmc", bmc.U_hat.shape, bmc.samples.shape)

    # ---- U2: orthogonalize on a synthetic 200x6 frame, keep 4
    rng = np.random.Generator(np.random.PCG64(17))
    truth = rng.standard_normal(200) * 3 + 10
    cols = {"N": np.arange(200), "Z": np.arange(200) % 17, "truth": truth}
    for j in range(6):
        cols[f"m{j}"] = truth + rng.normal(0.3 * j, 1.0, 200)
    df2 = pd.DataFrame(cols)
    models = [f"m{j}" for j in range(6)]
    b2 = BayesianModelCombination(models, {"BE": df2}, "truth")
    b2.orthogonalize("BE", df2.iloc[:150], 4)
    np.savez_compressed(
        os.path.join(HERE, "ortho_synth200x6.npz"),
        frame=df2[["N", "Z", "truth"] + models].to_numpy(float),
        U_hat=b2.U_hat, S_hat=b2.S_hat, Vt_hat=b2.Vt_hat,
        Vt_hat_normalized=b2.Vt_hat_normalized,
        centered_experiment_train=b2.centered_experiment_train,
        predictions_mean_train=b2._predictions_mean_train)
    print("ortho_synth200x6", b2.U_hat.shape)

    # ---- P1: posterior predictive + coverage, M=48 points, K_models=6, kept 4
    with pinned(71, 72):
        b2.train({"iterations": 12000, "sampler": "gibbs_sampling", "burn": 0,
                  "stepsize": 0.001, "b_mean_prior": np.zeros(4),
                  "b_mean_cov": np.diag(b2.S_hat ** 2), "nu0_chosen": 1.0,
                  "sigma20_chosen": 0.02})
    preds = df2.iloc[150:198][models].to_numpy()
    with pinned(81, 82):
        rndm_m, (lo, med, up) = rndm_m_random_calculator(preds, b2.samples, b2.Vt_hat)
    cov = coverage(np.arange(0, 101, 5), rndm_m, df2.iloc[150:198], "truth")
    np.savez_compressed(
        os.path.join(HERE, "predict_synth48.npz"),
        preds=preds, samples=b2.samples, Vt_hat=b2.Vt_hat, lower=lo, median=med,
        upper=up, coverage=np.array(cov), truth=df2.iloc[150:198]["truth"].to_numpy(),
        rndm_m_head=rndm_m[:64].copy(), rndm_m_sha=np.array(sha(rndm_m)),
        seed_g=np.int64(82), train_seed_z=np.int64(71), train_seed_g=np.int64(72))
    print("predict_synth48", rndm_m.shape, cov[:5])

    # ---- S1: simplex sampler on the reference's tiny case
    #          (tests/test_inference_utils.py:22-29), draws recorded via wrappers
    y = np.array([1.0, 2.0, 3.0])
    X = np.array([[1, 0], [0, 1], [1, 1]])
    Vt_hat = np.array([[0.5, 0.5], [0.5, -0.5]])
    S_hat = np.array([1.0, 0.5])
    with pinned(91, 92), recording_legacy() as log:
        s = gibbs_sampler_simplex(y, X, Vt_hat, S_hat, 200, [1.0, 1.0], burn=100,
                                  stepsize=0.01)
    shape = (1.0 + 3) / 2.0
    G = np.random.Generator(np.random.PCG64(92)).standard_gamma(shape, size=300)
    np.savez_compressed(
        os.path.join(HERE, "simplex_tiny3x2.npz"), samples=s, y=y, X=X, Vt_hat=Vt_hat,
        S_hat=S_hat, proposals=np.array([o for _, o in log["mvn"]]),
        prop_means=np.array([m for m, _ in log["mvn"]]),
        uniforms=np.array(log["unif"], float), G=G, burn=np.int64(100),
        stepsize=np.float64(0.01), nu0=np.float64(1.0), s20=np.float64(1.0))
    print("simplex_tiny3x2", s.shape, len(log["unif"]))

    # ---- S2: simplex on the 200x6 frame (inside-simplex start), 400 draws
    with pinned(93, 94), recording_legacy() as log:
        s = gibbs_sampler_simplex(b2.centered_experiment_train, b2.U_hat, b2.Vt_hat,
                                  b2.S_hat, 400, [1.0, 0.02], burn=200, stepsize=0.002)
    shape = (1.0 + 150) / 2.0
    G = np.random.Generator(np.random.PCG64(94)).standard_gamma(shape, size=600)
    np.savez_compressed(
        os.path.join(HERE, "simplex_synth150x4.npz"), samples=s,
        y=b2.centered_experiment_train, X=b2.U_hat, Vt_hat=b2.Vt_hat, S_hat=b2.S_hat,
        proposals=np.array([o for _, o in log["mvn"]]),
        uniforms=np.array(log["unif"], float), G=G, burn=np.int64(200),
        stepsize=np.float64(0.002), nu0=np.float64(1.0), s20=np.float64(0.02))
    print("simplex_synth150x4", s.shape, len(log["unif"]))


if __name__ == "__main__":
    main()
