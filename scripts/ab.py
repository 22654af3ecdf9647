#!/usr/bin/env python3
"""Same-box A/B of library builds, interleaved in ONE process (cdna guide 5.4 rule 24).

    python scripts/ab.py CASE[,CASE...] libA.so libB.so ... [--rounds R]

Builds come from `make -C pybmc_amd/csrc variant NAME=x EXTRA="-D..."` (-> .ab/lib_x.so); the
product library is pybmc_amd/libpybmc_amd.so.  Each round times every build once, in turn;
the table gives median and min of the loop kernel's HIP-event time per iteration (us).
Cases: c2 c2x8 c2x16 c4 c4x8 c5 c5x8 hbm hbmx8 n100k n100kx8 small small300 gram_c2/c4/c5 rss_c2/c4/c5/hbm
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pybmc_amd import _lib  # noqa: E402
from pybmc_amd.synthetic import synth_problem  # noqa: E402


def dense(n, k, dt, seed=8):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, k), dtype=np.float32)
    X *= np.float32(1.0 / np.sqrt(n))
    X = X.astype(dt, copy=False)
    y = (X @ rng.standard_normal(k).astype(dt) + 0.1 * rng.standard_normal(n)).astype(dt)
    return y, np.asfortranarray(X), (np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)


CASES = {   # name: (problem factory, dtype, chains, iterations)
    "c2": (lambda: synth(10000, 32), np.float64, 1, 20000),
    "c2x8": (lambda: synth(10000, 32), np.float64, 8, 20000),
    "c2x16": (lambda: synth(10000, 32), np.float64, 16, 20000),
    "c2x32": (lambda: synth(10000, 32), np.float64, 32, 20000),
    "c2x64": (lambda: synth(10000, 32), np.float64, 64, 10000),
    "c4": (lambda: dense(200000, 64, np.float32), np.float32, 1, 3000),
    "c4x8": (lambda: dense(200000, 64, np.float32), np.float32, 8, 1000),
    "c5": (lambda: dense(50000, 256, np.float64), np.float64, 1, 1500),
    "c5x8": (lambda: dense(50000, 256, np.float64), np.float64, 8, 600),
    "hbm": (lambda: dense(400000, 256, np.float32), np.float32, 1, 300),
    "hbmx8": (lambda: dense(400000, 256, np.float32), np.float32, 8, 150),
    "n100k": (lambda: dense(100000, 32, np.float64), np.float64, 1, 5000),
    "n100kx8": (lambda: dense(100000, 32, np.float64), np.float64, 8, 2000),
    "small": (lambda: synth(629, 3), np.float64, 1, 50000),
    # more chains that span XCDs (G > 32), register-resident
    "n30k64": (lambda: dense(30000, 64, np.float64), np.float64, 1, 5000),
    "n200k32": (lambda: dense(200000, 32, np.float64), np.float64, 1, 3000),
    "n100k64f32": (lambda: dense(100000, 64, np.float32), np.float32, 1, 4000),
    "n60k16": (lambda: dense(60000, 16, np.float64), np.float64, 1, 5000),
    # chains = 0: time the one-off Gram [X y]'[X y] (f64 MFMA) instead of the loop, us per launch
    "gram_c2": (lambda: synth(10000, 32), np.float64, 0, 50),
    "gram_c4": (lambda: dense(200000, 64, np.float32), np.float32, 0, 20),
    "gram_c5": (lambda: dense(50000, 256, np.float64), np.float64, 0, 20),
    "small300": (lambda: synth(629, 3), np.float64, 300, 20000),
    # chains = -1: time the stand-alone residual kernel (one coefficient vector), us per pass
    "rss_c4": (lambda: dense(200000, 64, np.float32), np.float32, -1, 200),
    "rss_c5": (lambda: dense(50000, 256, np.float64), np.float64, -1, 100),
    "rss_hbm": (lambda: dense(2000000, 64, np.float32), np.float32, -1, 30),
    "rss_c2": (lambda: synth(10000, 32), np.float64, -1, 200),
    # chains = -2: wall clock of bmc_set_prior (host K x K algebra + residual pass + rotation), ms
    "prior_c5": (lambda: dense(50000, 256, np.float64), np.float64, -2, 1),
    "prior_c4": (lambda: dense(200000, 64, np.float32), np.float32, -2, 1),
    "prior_c2": (lambda: synth(10000, 32), np.float64, -2, 1),
    # chains = -3: the simplex-constrained sampler (reference inference_utils.py:59-144), us per step
    "simplex_c2": (lambda: synth(10000, 32), np.float64, -3, 20000),
    "simplex_small": (lambda: synth(629, 3), np.float64, -3, 50000),
}


def synth(n, k):
    p = synth_problem(n, k + 1, k, seed=0)
    return p["y"], p["X"], p["prior"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases")
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    args = ap.parse_args()
    import ctypes
    _lib._share_hip_runtime_with_torch()
    libs = [(os.path.basename(p), _lib.bind(os.path.abspath(p), mode=ctypes.RTLD_LOCAL))
            for p in args.libs]
    for case in args.cases.split(","):
        make, dt, chains, iters = CASES[case]
        y, X, prior = make()
        ctxs = []
        for name, lib in libs:
            c = _lib.Context(0, lib=lib)
            c.set_problem(y, X, dtype=dt)
            c.set_prior(*prior)
            if args.groups or args.waves:
                c.set_tuning(args.groups, args.waves)
            if chains > 0:
                c.gibbs_run(chains, max(50, iters // 20), seeds=np.arange(chains) + 1)   # warm
            ctxs.append((name, c))
        times = {name: [] for name, _ in ctxs}
        first = None
        for r in range(args.rounds):
            # rotate the order: the first build timed after a pause gets a slightly higher clock
            order = ctxs[r % len(ctxs):] + ctxs[:r % len(ctxs)]
            for name, c in order:
                if chains == -3:
                    p_ = synth_problem(y.shape[0], X.shape[1] + 1, X.shape[1], seed=0)
                    Vt_hat = p_["Vt"] / p_["S_hat"][:, None]
                    out, acc, used, st = c.simplex_run(Vt_hat, p_["S_hat"], iters, 1.0, 0.02, 1000, 0.001,
                                                       seed=3, return_stats=True)
                    times[name].append(st["loop_ms"] * 1e3 / (iters + 1000))
                elif chains == -2:
                    import time
                    t0 = time.perf_counter()
                    c.set_prior(*prior)
                    times[name].append((time.perf_counter() - t0) * 1e3)
                    out, st = c.basis()[0], {"groups_per_chain": 0, "waves_per_group": 0,
                                             "chains_per_pass": 0, "residency": 0}
                elif chains <= 0:
                    times[name].append((c.gram_bench(reps=iters) if chains == 0
                                        else c.residual_rss_bench(nb=1, reps=iters)) * 1e3)
                    out, st = c.gram(), {"groups_per_chain": 0, "waves_per_group": 0,
                                         "chains_per_pass": 0, "residency": 0}
                else:
                    out, st = c.gibbs_run(chains, iters, seeds=np.arange(chains) + 1)
                    times[name].append(st["loop_ms"] * 1e3 / iters)
                if first is None:
                    first = out
                elif not np.array_equal(out, first):
                    print(f"  NOTE {case}: {name} differs from the first build by "
                          f"{np.abs(out - first).max():.3e}", flush=True)
        for name, _ in ctxs:
            t = np.array(times[name])
            print(f"{case:9s} {name:28s} median {np.median(t):8.4f}  min {t.min():8.4f}  "
                  f"max {t.max():8.4f} us/iter   ({chains} chains, geometry G={st['groups_per_chain']} "
                  f"W={st['waves_per_group']} cpp={st['chains_per_pass']} res={st['residency']})",
                  flush=True)
        for _, c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
