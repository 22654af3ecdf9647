"""Dev: C2 geometry sweep (product build, one process, interleaved rounds)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 20000
cfgs = [(32, 5, 1), (27, 6, 1), (23, 7, 1), (20, 8, 1), (32, 3, 2), (27, 3, 2), (20, 4, 2), (16, 5, 2), (14, 6, 2), (10, 8, 2)]
res = {c: [] for c in cfgs}
for rnd in range(3):
    for c in cfgs:
        ctx.set_tuning(groups_per_chain=c[0], waves_per_group=c[1], residency=1, panels_per_wave=c[2])
        _, st = ctx.gibbs_run(1, T, seeds=[1])
        assert st["residency"] == 1 and st["groups_per_chain"] == c[0], (c, st)
        res[c].append(st["loop_ms"] * 1e3 / T)
for c, v in res.items():
    print(c, f"median {np.median(v):.3f} us/iter")
