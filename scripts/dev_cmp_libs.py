"""Dev: time C2 single chain with a given library file (argv[1]) -- run per library."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 20000
ctx.gibbs_run(1, 2000, seeds=[1])
v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(5)]
print(sys.argv[1].split("/")[-1], "G", v[0]["groups_per_chain"], "W", v[0]["waves_per_group"], "us/iter", sorted(round(s["loop_ms"] * 1e3 / T, 3) for s in v))
