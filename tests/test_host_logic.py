"""Host-side logic that needs no GPU: truncation, centring, chain partition, seeds,
constructor validation, coverage arithmetic.  Mirrors the reference's own
tests/test_inference_utils.py:34-45 and tests/test_bmc.py:34-81."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from pybmc_amd import BayesianModelCombination, USVt_hat_extraction, coverage
from pybmc_amd.chains import chain_block, chain_seeds, max_block, posterior_summary


def test_usvt_hat_extraction_shapes():  # reference tests/test_inference_utils.py:34-45
    U = np.array([[1, 0], [0, 1]])
    S = np.array([2.0, 1.0])
    Vt = np.array([[1, 0], [0, 1]])
    U_hat, S_hat, Vt_hat, Vt_n = USVt_hat_extraction(U, S, Vt, 2)
    assert U_hat.shape == (2, 2) and len(S_hat) == 2
    assert Vt_hat.shape == (2, 2) and Vt_n.shape == (2, 2)
    assert np.array_equal(Vt_hat, np.array([[0.5, 0.0], [0.0, 1.0]]))


def test_usvt_hat_extraction_matches_reference_values():
    g = load_golden("ortho_synth200x6")
    fr = g["frame"]
    F, truth = fr[:150, 3:], fr[:150, 2]
    Fc = F - F.mean(axis=1)[:, None]
    U, S, Vt = np.linalg.svd(Fc)  # what the reference feeds it (bmc.py:119)
    U_hat, S_hat, Vt_hat, Vt_n = USVt_hat_extraction(U, S, Vt, 4)
    assert np.array_equal(U_hat, g["U_hat"]) and U_hat.flags.f_contiguous
    assert np.array_equal(S_hat, g["S_hat"]) and np.array_equal(Vt_n, g["Vt_hat_normalized"])
    assert np.array_equal(Vt_hat, g["Vt_hat"])


def make_bmc():
    df = pd.DataFrame({
        "x": [1, 2, 3, 4, 5, 6], "y": [10, 11, 12, 13, 14, 15],
        "truth": [11, 21, 31, 41, 51, 61], "model1": [10, 20, 30, 40, 50, 60],
        "model2": [15, 25, 35, 45, 55, 65], "model3": [12, 30, 32, 43, 58, 67]})
    bmc = BayesianModelCombination(["model1", "model2", "model3", "truth"], {"target": df}, "truth")
    return bmc, df


def test_bmc_init_validation():  # reference tests/test_bmc.py:34-52
    data = {"property": pd.DataFrame({"model1": [1, 2], "model2": [3, 4]})}
    b = BayesianModelCombination(["model1", "model2"], data, "truth")
    assert b.models_list == ["model1", "model2"] and b.truth_column_name == "truth"
    assert b.samples is None and b.Vt_hat is None
    with pytest.raises(ValueError):
        BayesianModelCombination("not_a_list", data, "truth")
    with pytest.raises(ValueError):
        BayesianModelCombination(["model1"], "not_a_dict", "truth")


def test_orthogonalize_matches_reference():  # reference tests/test_bmc.py:54-81 + golden values
    bmc, df = make_bmc()
    bmc.orthogonalize("target", df.iloc[:4], 2)
    g = load_golden("ortho_testbmc")
    assert bmc.U_hat.shape[0] == 4 and bmc.Vt_hat.shape[1] == len(bmc.models)
    assert np.array_equal(bmc.centered_experiment_train, g["centered_experiment_train"])
    assert np.array_equal(bmc._predictions_mean_train, g["predictions_mean_train"])
    # thin vs full SVD: equal up to rounding, same signs
    np.testing.assert_allclose(bmc.U_hat, g["U_hat"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(bmc.S_hat, g["S_hat"], rtol=1e-13)
    np.testing.assert_allclose(bmc.Vt_hat, g["Vt_hat"], rtol=0, atol=1e-13)
    assert bmc.current_property == "target"


def test_orthogonalize_rejects_null_space():  # quirk Q10
    bmc, df = make_bmc()
    with pytest.raises(ValueError):
        bmc.orthogonalize("target", df.iloc[:4], 3)


def test_predict_before_train_raises_value_error():  # quirk Q7
    bmc, df = make_bmc()
    with pytest.raises(ValueError):
        bmc.predict(df)
    with pytest.raises(ValueError):
        bmc.predict2("target")
    with pytest.raises(ValueError):
        bmc.train()


def test_coverage_matches_reference():
    g = load_golden("predict_synth48")
    # rebuild a frame that holds the truth column only
    df = pd.DataFrame({"truth": g["truth"]})
    # coverage needs the full draw matrix; its head + the stored result pin the arithmetic
    rng = np.random.Generator(np.random.PCG64(3))
    rndm = rng.standard_normal((10000, 48)) * 2 + g["truth"][None, :]
    got = coverage(np.arange(0, 101, 5), rndm, df, "truth")
    srt = np.sort(rndm, axis=0)
    want = []
    for p in np.arange(0, 101, 5):
        lo, hi = int((0.5 - p / 200) * 10000), int((0.5 + p / 200) * 10000) - 1
        want.append(sum(srt[lo, i] <= g["truth"][i] <= srt[hi, i] for i in range(48)) / 48 * 100)
    assert got == want
    assert got[0] == 0.0  # p = 0 can never cover (lower index above upper index)


def test_chain_partition_and_seeds():
    for n, w in [(8, 1), (8, 2), (8, 8), (3, 2), (5, 4), (0, 2), (1, 4)]:
        blocks = [chain_block(n, w, r) for r in range(w)]
        assert sum(blocks, []) == list(range(n))
        assert max(len(b) for b in blocks) == (max_block(n, w) if n else 0)
        assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1
    s_all = chain_seeds(7, list(range(8)))
    assert len(set(s_all.tolist())) == 8
    # a chain's seed depends on its global id only, not on the partition
    for w in (1, 2, 4, 8):
        got = np.concatenate([chain_seeds(7, chain_block(8, w, r)) for r in range(w)])
        assert np.array_equal(got, s_all)
    with pytest.raises(ValueError):
        chain_block(4, 2, 2)


def test_posterior_summary_weights_sum_to_one():
    g = load_golden("gibbs_ortho629x3")
    Vt_hat = g["Vt"] / g["S_hat"][:, None]
    s = posterior_summary(g["samples"], Vt_hat)
    assert abs(s["weights_mean"].sum() - 1.0) < 1e-12
    assert s["beta_mean"].shape == (3,)


def test_bench_cpu_baseline_leg_runs_on_the_host():
    """bench.py's cpu_baseline (the numpy port of the reference loop, the only thing in bench.py
    that may use oracle/) on a tiny problem and a tiny time budget: the fields the judge reads."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from pybmc_amd.synthetic import synth_problem
    p = synth_problem(400, 5, 4, 0)
    out = bench.cpu_baseline(p, budget_s=0.3, chunk=50)
    assert out["kind"] == "port" and out["unit"] == "samples/s" and out["value"] > 0
    assert out["cores"] >= 1 and "iterations" in out["sample"]
    assert out.get("value_1_thread", 1.0) > 0


def test_run_on_devices_splits_chains_like_ranks(monkeypatch):
    """train(devices=[...]) -> chains.run_on_devices: one context and one host thread per
    device, chains split like ranks split them, blocks concatenated in global chain order, a
    device listed twice used once.  The contexts are stand-ins (no GPU here)."""
    import threading
    from pybmc_amd import _lib, chains

    made, lock = {}, threading.Lock()

    class FakeCtx:
        def __init__(self, dev):
            self.dev, self.threads, self.problem, self.prior = dev, set(), None, None

        def set_problem(self, y, X, dtype=None):
            self.problem = (len(y), X.shape, dtype)

        def set_prior(self, b0, C0, nu0, s20):
            self.prior = (nu0, s20)

        def gibbs_run(self, n_chains, iters, seeds=None):
            with lock:
                self.threads.add(threading.get_ident())
            out = np.empty((n_chains, iters, 3))
            for c, s in enumerate(seeds):
                out[c] = float(s) + 1000.0 * self.dev
            return out, {"device": self.dev, "n_chains": n_chains}

    def fake_default_context(dev=0):
        with lock:
            return made.setdefault(dev, FakeCtx(dev))

    monkeypatch.setattr(_lib, "default_context", fake_default_context)
    y, X = np.zeros(10), np.zeros((10, 2))
    prior = (np.zeros(2), np.eye(2), 1.0, 0.02)
    seeds = np.arange(1, 8, dtype=np.uint64)
    out, stats = chains.run_on_devices(y, X, 5, prior, 7, seeds, [0, 1, 1, 2], dtype=np.float32)
    assert out.shape == (7, 5, 3) and sorted(made) == [0, 1, 2]
    # chain_block(7, 3, r): [0,1,2], [3,4], [5,6] -> devices 0, 1, 2
    want_dev = [0, 0, 0, 1, 1, 2, 2]
    assert np.array_equal(out[:, 0, 0], seeds.astype(float) + 1000.0 * np.array(want_dev))
    assert [s["n_chains"] for s in stats] == [3, 2, 2]
    assert all(c.problem == (10, (10, 2), np.float32) and c.prior == (1.0, 0.02) for c in made.values())
    main = threading.get_ident()                 # each context driven by one worker thread
    assert all(len(c.threads) == 1 and main not in c.threads for c in made.values())
    one, _ = chains.run_on_devices(y, X, 5, prior, 1, [9], [2])
    assert one.shape == (5, 3) and one[0, 0] == 9.0 + 2000.0
    with pytest.raises(ValueError):
        chains.run_on_devices(y, X, 5, prior, 2, [1, 2], [])

    class Boom(FakeCtx):
        def gibbs_run(self, *a, **k):
            raise np.linalg.LinAlgError("Singular matrix")
    made[1] = Boom(1)
    with pytest.raises(np.linalg.LinAlgError):      # a worker's exception reaches the caller
        chains.run_on_devices(y, X, 5, prior, 7, seeds, [0, 1, 2])
