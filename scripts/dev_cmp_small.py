"""Dev: notebook-sized chain in ONE workgroup (no exchange), a given library file."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
T = 20000
p = synth_problem(629, 4, 3, 3)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
for tune in (dict(groups_per_chain=1, waves_per_group=4), dict(groups_per_chain=1, waves_per_group=8), dict(groups_per_chain=10, waves_per_group=1)):
    ctx.set_tuning(**tune)
    ctx.gibbs_run(1, 2000, seeds=[1])
    v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(5)]
    print(sys.argv[1].split("/")[-1], tune, "G", v[0]["groups_per_chain"], "W", v[0]["waves_per_group"], "res", v[0]["residency"], "us/iter", sorted(round(s["loop_ms"] * 1e3 / T, 3) for s in v))
