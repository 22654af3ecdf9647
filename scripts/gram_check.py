#!/usr/bin/env python3
"""Gram kernel ([X y]'[X y], f64 MFMA) against numpy over a sweep of shapes (every tile-count
class of gram_geometry, f32 and f64 storage, ragged row counts), and its time per launch.
Measurement / diagnostic only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from pybmc_amd import _lib  # noqa: E402


def main():
    ctx = _lib.Context(0)
    rng = np.random.Generator(np.random.PCG64(3))
    shapes = [(3, 2), (629, 3), (1237, 5), (10000, 32), (5000, 15), (4097, 47), (3000, 63), (200000, 64),
              (7000, 79), (6000, 95), (9000, 130), (5000, 143), (5000, 144), (8000, 160), (4000, 200),
              (50000, 256), (2000, 255)]
    if len(sys.argv) > 1:   # e.g. "50000x256,200000x64": only these shapes
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
    worst = 0.0
    for n, k in shapes:
        for dt in (np.float64, np.float32):
            X = rng.standard_normal((n, k)).astype(dt)
            y = rng.standard_normal(n).astype(dt)
            ctx.set_problem(y, np.asfortranarray(X), dtype=dt)
            Xa = np.column_stack([X.astype(np.float64), y.astype(np.float64)])
            want = Xa.T @ Xa
            got = ctx.gram()
            err = np.abs(got - want).max() / np.abs(want).max()
            worst = max(worst, err)
            ms = ctx.gram_bench(reps=10) if n >= 5000 else float("nan")
            fl = 2.0 * n * (k + 1) ** 2
            print(f"n={n:7d} k={k:3d} {np.dtype(dt).name:8s} rel err {err:.2e}  {ms * 1e3:8.1f} us"
                  f"  {fl / (ms * 1e-3) / 1e12 if ms == ms else 0:6.2f} TF (full square)", flush=True)
            assert err < 1e-12, (n, k, dt, err)
            assert np.array_equal(got, got.T)
    print("worst rel err", worst)


if __name__ == "__main__":
    main()
