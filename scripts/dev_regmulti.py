"""Dev: register-residency chains-per-pass -- geometry, parity with solo runs, and timing."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
ctx = _lib.Context(0)
T = 3000
for n, k, dt, nch in ((100000, 32, np.float64, 8), (120000, 7, np.float64, 5), (200000, 64, np.float32, 8),
                      (150000, 20, np.float32, 4), (200000, 64, np.float64, 8), (60000, 16, np.float64, 8)):
    rng = np.random.default_rng(n + k)
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(np.float64) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    ctx.set_problem(y, np.asfortranarray(X), dtype=dt)
    ctx.set_prior(np.zeros(k), np.eye(k) * 10.0, 1.0, 0.02)
    seeds = np.arange(nch) + 11
    ctx.set_tuning(chains_per_pass=1)
    ctx.gibbs_run(nch, 200, seeds=seeds)
    solo, s1 = ctx.gibbs_run(nch, T, seeds=seeds)
    ctx.set_tuning()
    ctx.gibbs_run(nch, 200, seeds=seeds)
    sh, s2 = ctx.gibbs_run(nch, T, seeds=seeds)
    err = np.abs(sh - solo).max()
    keys = ("residency", "groups_per_chain", "waves_per_group", "chains_per_pass", "launches")
    print(f"N={n} K={k} {np.dtype(dt).name} chains={nch}: solo", {q: s1[q] for q in keys},
          f"{s1['loop_ms'] * 1e3 / T:.2f} us/iter(all chains) | shared", {q: s2[q] for q in keys},
          f"{s2['loop_ms'] * 1e3 / T:.2f} us/iter  maxdiff {err:.2e}", flush=True)
