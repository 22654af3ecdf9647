"""Dev: single-workgroup chain at the reference's own size (N=629, K=3): waves x panels per wave."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(629, 4, 3, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 20000
for W, ppw in ((0, 0), (3, 4), (5, 2), (6, 2), (8, 2), (8, 1)):
    try:
        ctx.set_tuning(groups_per_chain=1 if W else 0, waves_per_group=W, panels_per_wave=ppw)
        ctx.gibbs_run(1, 1000, seeds=[1])
        v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(5)]
        print((W, ppw), "G", v[0]["groups_per_chain"], "W", v[0]["waves_per_group"], sorted(round(s["loop_ms"] * 1e3 / T, 3) for s in v), flush=True)
    except Exception as e:
        print((W, ppw), "->", e, flush=True)
ctx.set_tuning()
