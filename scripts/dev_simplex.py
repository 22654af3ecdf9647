"""Dev: timing of the simplex-constrained sampler at the C2 size and at the notebook size."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
for n, km, k in ((10000, 33, 32), (629, 4, 3)):
    p = synth_problem(n, km, k, 0)
    ctx.set_problem(p["y"], p["X"])
    T = 20000
    ctx.simplex_run(p["Vt"] / p["S_hat"][:, None], p["S_hat"], 500, 1.0, 0.02, 100, 0.001, seed=1)
    out, acc, used, st = ctx.simplex_run(p["Vt"] / p["S_hat"][:, None], p["S_hat"], T, 1.0, 0.02, 2000, 0.001, seed=1, return_stats=True)
    tot = T + 2000
    print(f"simplex N={n} K={k}: G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} {st['loop_ms']*1e3/tot:.3f} us/iter, accepted {acc}/{T}, uniforms {used}", flush=True)
