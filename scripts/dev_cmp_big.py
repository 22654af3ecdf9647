"""Dev: time the whole-chip configurations (C4, C5, N=100000 K=32) with a given library file
(argv[1]) -- run once per library for a same-box A/B of two builds."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
for name, n, k, dt, T in (("C4 200000x64 f32", 200000, 64, np.float32, 4000), ("100000x32 f64", 100000, 32, np.float64, 4000),
                          ("C5 50000x256 f64", 50000, 256, np.float64, 1500)):
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(float) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    ctx.set_problem(y, np.asfortranarray(X), dtype=dt); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
    ctx.gibbs_run(1, 300, seeds=[1])
    v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(4)]
    print(sys.argv[1].split("/")[-1], name, "G", v[0]["groups_per_chain"], "W", v[0]["waves_per_group"], "res", v[0]["residency"],
          "local", v[0]["xcd_local_chains"], "us/iter", sorted(round(s["loop_ms"] * 1e3 / T, 3) for s in v), flush=True)
