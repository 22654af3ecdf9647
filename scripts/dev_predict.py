"""Dev: predictive path timing (reference: M=5000, K=32 -> 11.7 s on CPU)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
import os
_lib.LIB_PATH = os.environ.get('BMC_LIB', _lib.LIB_PATH)
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
for M, Km, k in ((5000, 32, 31), (50000, 257, 256)):
    preds = rng.standard_normal((M, Km)) + 5
    theta = np.column_stack([rng.standard_normal((10000, k)) * 0.01, rng.uniform(0.5, 1.5, 10000)])
    Vt = rng.standard_normal((k, Km))
    truth = rng.standard_normal(M) + 5
    for want in (False, True):
        ctx.predict(preds[:64], theta, Vt, seed=1, want_draws=False)
        t0 = time.time()
        r, bands, cov = ctx.predict(preds, theta, Vt, seed=1, truth=truth, cov_percentiles=list(range(0, 101, 5)), want_draws=want)
        dt = time.time() - t0
        print(f"M={M} Km={Km}: predict {'with' if want else 'without'} rndm_m copy-back: {dt:.3f} s  (draws {10000*M*8/1e9:.2f} GB) cov[10]={cov[10]:.1f}")
