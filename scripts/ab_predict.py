#!/usr/bin/env python3
"""Same-box A/B of library builds on the C5 posterior predictive (M = 50000 points, 257 models,
10000 draws): HIP-event times of the GEMM and the order statistics, interleaved in one process;
checks that the bands agree bit for bit.   python scripts/ab_predict.py libA.so libB.so ..."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pybmc_amd import _lib  # noqa: E402


def main():
    _lib._share_hip_runtime_with_torch()
    libs = [(os.path.basename(p), _lib.bind(os.path.abspath(p), mode=ctypes.RTLD_LOCAL)) for p in sys.argv[1:]]
    rng = np.random.Generator(np.random.PCG64(55))
    M, Km, k, S = 50000, 257, 256, 10000
    preds = rng.standard_normal((M, Km))
    Vt = rng.standard_normal((k, Km)) * 0.05
    theta = np.column_stack([rng.standard_normal((S, k)) * 0.1, rng.uniform(0.05, 0.15, S)])
    ctxs = [(n, _lib.Context(0, lib=l)) for n, l in libs]
    times = {n: [] for n, _ in ctxs}
    first = None
    for r in range(5):
        for n, c in ctxs[r % len(ctxs):] + ctxs[:r % len(ctxs)]:
            res = c.predict(preds, theta, Vt, seed=9, truth=preds.mean(1),
                            cov_percentiles=list(range(0, 101, 5)), want_draws=False)
            t = c.predict_timing()
            times[n].append((t["gemm_ms"], t["select_ms"]))
            bands = np.asarray(res[1])
            if first is None:
                first = bands
            elif not np.array_equal(bands, first):
                print(f"  NOTE: {n} bands differ by {np.abs(bands - first).max():.3e}", flush=True)
    for n, _ in ctxs:
        t = np.array(times[n][1:])
        print(f"{n:24s} gemm median {np.median(t[:, 0]):7.3f} ms (min {t[:, 0].min():7.3f})   order statistics "
              f"{np.median(t[:, 1]):6.3f} ms", flush=True)


if __name__ == "__main__":
    main()
