#!/usr/bin/env python3
"""The kernels bench.py's headline step does not launch, run once each at the BASELINE sizes so
that one rocprofv3 pass set (scripts/profile_round.sh) sees them: several chains per pass
(gibbs_multi_kernel) at C4 / C5 / 410 MB, the streaming loop, the residual and Gram kernels at
C4 / C5, the posterior predictive at C5 (M = 50000, 257 models, 10000 draws) and the simplex
sampler at the C2 size, and the one-wave kernels at the reference's own size (C1: N = 629,
3 components).  Measurement workload only; prints one line per case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: one HIP runtime in the process)
from pybmc_amd import _lib  # noqa: E402
from pybmc_amd.synthetic import synth_problem  # noqa: E402


def dense(n, k, dt, seed=8):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, k), dtype=np.float32)
    X *= np.float32(1.0 / np.sqrt(n))
    X = X.astype(dt, copy=False)
    y = (X @ rng.standard_normal(k).astype(dt) + 0.1 * rng.standard_normal(n)).astype(dt)
    return y, np.asfortranarray(X), (np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)


def main():
    ctx = _lib.Context(0)
    # C2: 8 and 16 chains in one launch (one / two chains per XCD), 32 and 64 as bundles of 4 / 8
    # chains per XCD (gibbs_multi_kernel with bundle slots)
    p = synth_problem(10000, 33, 32, seed=0)
    ctx.set_problem(p["y"], p["X"])
    ctx.set_prior(*p["prior"])
    for c in (8, 16, 32, 64):
        _, st = ctx.gibbs_run(c, 5000, seeds=np.arange(c) + 1)
        print(f"c2 x{c}: {st['loop_ms'] * 1e3 / 5000:.3f} us/iter", flush=True)
    # simplex sampler on the same problem (reference inference_utils.py:59-144)
    Vt_hat = p["Vt"] / p["S_hat"][:, None]
    _, acc, used, st = ctx.simplex_run(Vt_hat, p["S_hat"], 4000, 1.0, 0.02, 1000, 0.001, seed=3,
                                       return_stats=True)
    print(f"simplex c2: {st['loop_ms'] * 1e3 / 5000:.3f} us/iter, accepted {acc}", flush=True)
    # C1, the reference's own size: one wave per chain (gibbs_wave_kernel / simplex_wave_kernel)
    p1 = synth_problem(629, 4, 3, seed=0)
    ctx.set_problem(p1["y"], p1["X"])
    ctx.set_prior(*p1["prior"])
    for c in (1, 256):
        _, st = ctx.gibbs_run(c, 20000, seeds=np.arange(c) + 1)
        print(f"c1 x{c}: {st['loop_ms'] * 1e3 / 20000:.3f} us/iter (waves {st['waves_per_group']})", flush=True)
    _, acc, used, st = ctx.simplex_run(p1["Vt"] / p1["S_hat"][:, None], p1["S_hat"], 20000, 1.0, 0.02, 1000,
                                       0.001, seed=3, return_stats=True)
    print(f"simplex c1: {st['loop_ms'] * 1e3 / 21000:.3f} us/step, accepted {acc}", flush=True)
    # a few thousand rows: the same kernel in 4 waves per chain
    yq, Xq, priorq = dense(2500, 8, np.float64)
    ctx.set_problem(yq, Xq)
    ctx.set_prior(*priorq)
    for c in (1, 256):
        _, st = ctx.gibbs_run(c, 10000, seeds=np.arange(c) + 1)
        print(f"n2500k8 x{c}: {st['loop_ms'] * 1e3 / 10000:.3f} us/iter (waves {st['waves_per_group']})", flush=True)
    for tag, n, k, dt, it1, it8 in (("c4", 200000, 64, np.float32, 2000, 500),
                                    ("c5", 50000, 256, np.float64, 1000, 300),
                                    ("hbm", 400000, 256, np.float32, 200, 60)):
        y, X, prior = dense(n, k, dt)
        ctx.set_problem(y, X, dtype=dt)
        ctx.set_prior(*prior)
        _, st1 = ctx.gibbs_run(1, it1, seeds=[1])
        _, st8 = ctx.gibbs_run(8, it8, seeds=np.arange(8) + 1)
        ms_r = ctx.residual_rss_bench(nb=1, reps=20)
        ms_g = ctx.gram_bench(reps=10)
        print(f"{tag}: loop {st1['loop_ms'] * 1e3 / it1:.2f} us/iter, x8 {st8['loop_ms'] * 1e3 / it8:.2f} "
              f"(cpp {st8['chains_per_pass']}), residual {ms_r * 1e3:.1f} us, gram {ms_g * 1e3:.1f} us",
              flush=True)
    # posterior predictive at the C5 size
    rng = np.random.Generator(np.random.PCG64(55))
    M, Km, k, S = 50000, 257, 256, 10000
    preds = rng.standard_normal((M, Km))
    Vt = rng.standard_normal((k, Km)) * 0.05
    theta = np.column_stack([rng.standard_normal((S, k)) * 0.1, rng.uniform(0.05, 0.15, S)])
    for _ in range(3):
        ctx.predict(preds, theta, Vt, seed=9, truth=preds.mean(1),
                    cov_percentiles=list(range(0, 101, 5)), want_draws=False)
    print("predict c5:", ctx.predict_timing(), flush=True)
    # and with the draws returned in the reference's layout (device transpose + staged copy)
    ctx.predict(preds[:8192], theta, Vt, seed=9, want_draws=True)


if __name__ == "__main__":
    main()
