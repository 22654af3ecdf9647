"""Dev: time C2 (1 chain) and a notebook-sized problem with a given library file (argv[1]) --
run once per library for a same-box A/B of two builds."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
T = 20000
for n, km, k in ((10000, 33, 32), (629, 4, 3)):
    p = synth_problem(n, km, k, 0)
    ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
    ctx.gibbs_run(1, 2000, seeds=[1])
    v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(5)]
    o8, s8 = ctx.gibbs_run(8, T, seeds=np.arange(8) + 1)
    print(sys.argv[1].split("/")[-1], f"N={n} K={k}", "G", v[0]["groups_per_chain"], "W", v[0]["waves_per_group"], "us/iter",
          sorted(round(s["loop_ms"] * 1e3 / T, 3) for s in v), "8 chains:", round(s8["loop_ms"] * 1e3 / T, 3))
