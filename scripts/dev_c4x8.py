"""Dev: C4 (N=200000, K=64, f32) with 8 chains: register residency (4 chains per pass) against
streaming (8 chains per pass)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
n, k, dt = 200000, 64, np.float32
X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
y = (X.astype(float) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
ctx.set_problem(y, np.asfortranarray(X), dtype=dt); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
T = 2000
for res in (0, 1, 3):
    for C in (1, 8):
        ctx.set_tuning(residency=res)
        ctx.gibbs_run(C, 200, seeds=np.arange(C) + 1)
        out, st = ctx.gibbs_run(C, T, seeds=np.arange(C) + 1)
        print(f"residency request {res}: chains={C} -> res {st['residency']} G {st['groups_per_chain']} W {st['waves_per_group']} cpp {st['chains_per_pass']} launches {st['launches']}: {st['loop_ms']*1e3/T:.2f} us/iter(all)", flush=True)
ctx.set_tuning()
