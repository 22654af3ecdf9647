"""Dev: does the time per iteration depend on the chain length? (C2, one chain)"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
ctx.gibbs_run(1, 2000, seeds=[1])
for T in (5000, 20000, 50000, 100000, 20000, 50000):
    v = [ctx.gibbs_run(1, T, seeds=[1])[1] for _ in range(3)]
    print(T, sorted(round(s["loop_ms"] * 1e3 / T, 4) for s in v), flush=True)
