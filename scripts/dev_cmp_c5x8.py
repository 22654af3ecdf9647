"""Dev: C5 with 1 and 8 chains with a given library file (argv[1])."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
n, k, dt = 50000, 256, np.float64
X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
y = (X @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
ctx.set_problem(y, np.asfortranarray(X)); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
ctx.gibbs_run(1, 200, seeds=[1])
for C in (1, 8):
    v = [ctx.gibbs_run(C, 600, seeds=np.arange(C) + 1)[1]["loop_ms"] / 600 * 1e3 for _ in range(3)]
    print(sys.argv[1].split("/")[-1], "C5", C, "chains", [round(x, 2) for x in v], flush=True)
